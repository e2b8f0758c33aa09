"""Shared builders for the parity tests: the graphs behind tests/golden/*.npz."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TYPES = ['t1', 't2', 't3']


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def readme_graph():
    """reference README.md:50-64 (BASELINE config 1)."""
    R = {('t1', 't2'): [np.random.RandomState(0).rand(50, 100)],
         ('t1', 't3'): [np.random.RandomState(1).rand(50, 40)],
         ('t2', 't3'): [np.random.RandomState(2).rand(100, 40)]}
    return R, list(TYPES), {'t1': 10, 't2': 20, 't3': 30}


def probe_graph(z):
    """multi-relation / negative relation / Theta / masks graph stored in probe_multirel.npz."""
    R = {('t1', 't2'): [z['R_t1_t2_0'], z['R_t1_t2_1']],
         ('t1', 't3'): [z['R_t1_t3_0']],
         ('t2', 't3'): [z['R_t2_t3_0']]}
    Theta = {('t1', 't1'): [z['Theta_t1_0']], ('t2', 't2'): [z['Theta_t2_0'], z['Theta_t2_1']]}
    M = {('t1', 't2'): [z['M_t1_t2_0'], None], ('t1', 't3'): [None], ('t2', 't3'): [z['M_t2_t3_0']]}
    return R, Theta, M, list(TYPES), {'t1': 6, 't2': 5, 't3': 4}


def rank_deficient_graph(z):
    R = {('t1', 't2'): [z['R_t1_t2_0']], ('t1', 't3'): [z['R_t1_t3_0']]}
    return R, list(TYPES), {'t1': 50, 't2': 30, 't3': 10}


def dicty_graph():
    """BASELINE config 2 inputs (values of reference datasets/base.py:47-61, as a data fixture)."""
    z = golden('dicty_inputs.npz')
    shp = tuple(z['ann_shape'])
    ann = np.unpackbits(z['ann_bits'], axis=1)[:, :shp[1]].astype(np.float64)
    x = z['expr_milli'].astype(np.int64) / 1000.0
    expr = np.log(np.maximum(x, np.finfo(float).eps))          # datasets/base.py:57
    ppi = np.zeros(tuple(z['ppi_shape']))
    ppi[z['ppi_rows'], z['ppi_cols']] = z['ppi_vals']
    R = {('gene', 'go'): [ann], ('gene', 'exc'): [expr]}
    Theta = {('gene', 'gene'): [ppi]}
    return R, Theta, ['gene', 'go', 'exc'], {'gene': 50, 'go': 15, 'exc': 5}


def c3_scaled_graph(z):
    from oracle.dfmf_oracle import hash_uniform_matrix
    n1, n2, n3 = [int(v) for v in z['shape']]
    c = [int(v) for v in z['ranks']]
    ds, gs = z['data_seeds'], z['g0_seeds']
    R = {('t1', 't2'): [hash_uniform_matrix(int(ds[0]), n1, n2)],
         ('t1', 't3'): [hash_uniform_matrix(int(ds[1]), n1, n3)],
         ('t2', 't3'): [hash_uniform_matrix(int(ds[2]), n2, n3)]}
    G0 = {('t1', 't1'): hash_uniform_matrix(int(gs[0]), n1, c[0]),
          ('t2', 't2'): hash_uniform_matrix(int(gs[1]), n2, c[1]),
          ('t3', 't3'): hash_uniform_matrix(int(gs[2]), n3, c[2])}
    return R, G0, list(TYPES), dict(zip(TYPES, c))


def c3_planted_graph(shape=(2000, 4000, 1600), ranks=(128, 256, 256), noise=0.01, bf16=False):
    """The PLANTED variant of BASELINE config 3: R_ij = G*_i S*_ij G*_j^T / mean + noise * U with G*, S*, U
    from the counter-based generator (seeds 200 + type, 300 + relation, 400 + relation), G0 from seeds 100 + type -- the one
    workload on which the RMSE discriminates (iid-uniform data sits at sqrt(1/12) whatever the fit); noise floor
    noise / sqrt(12).  bf16: the relations rounded to bf16 (what the SKF_BF16 engine stores).  1/25 linear scale by default."""
    from oracle.dfmf_oracle import hash_uniform_matrix
    n = dict(zip(TYPES, shape))
    c = dict(zip(TYPES, ranks))
    Gs = {t: hash_uniform_matrix(200 + q, n[t], c[t]) for q, t in enumerate(TYPES)}
    R = {}
    for seed, (i, j) in enumerate([('t1', 't2'), ('t1', 't3'), ('t2', 't3')]):
        Rm = Gs[i].dot(hash_uniform_matrix(300 + seed, c[i], c[j])).dot(Gs[j].T)
        Rm /= Rm.mean()
        Rm += noise * hash_uniform_matrix(400 + seed, n[i], n[j])
        if bf16:
            import skfusion_amd._native as nat
            Rm = nat.from_bf16_bits(nat.to_bf16_bits(Rm)).astype(np.float64)
        R[i, j] = [Rm]
    G0 = {(t, t): hash_uniform_matrix(100 + q, n[t], c[t]) for q, t in enumerate(TYPES)}
    return R, G0, list(TYPES), c


C5_TYPES = ['user', 'movie', 'genre', 'actor', 'tag', 'director']
C5_SIZES = {'user': 400, 'movie': 240, 'genre': 16, 'actor': 200, 'tag': 120, 'director': 80}
C5_RANKS = {'user': 16, 'movie': 24, 'genre': 6, 'actor': 12, 'tag': 8, 'director': 8}
# (row, col, data seed, density of the binary relation | None for the ratings)
C5_RELATIONS = [('user', 'movie', 50, None), ('movie', 'genre', 51, 0.15), ('movie', 'actor', 52, 0.03),
                ('movie', 'tag', 53, 0.05), ('movie', 'director', 54, 0.02), ('user', 'tag', 55, 0.04)]


def movielens_style_graph(sizes=None, ranks=None, masked=0.98, lam=0.01):
    """BASELINE config 5, scaled down: a MovieLens-style graph in the shape of reference
    examples/movielens_completion.py:22-79 -- 6 object types, 6 relations, the ratings relation
    User x Movie (values {0.5..5}/5) with `masked` of its entries unknown (M = True), binary
    side relations, Theta_user = lam*I, Theta_movie = [lam*I, sparse negative similarity].
    Every entry derives from the counter-based generator shared with the device fill kernel."""
    from oracle.dfmf_oracle import hash_uniform_matrix
    n = dict(sizes or C5_SIZES)
    c = dict(ranks or C5_RANKS)
    R, M = {}, {}
    for i, j, seed, dens in C5_RELATIONS:
        u = hash_uniform_matrix(seed, n[i], n[j])
        if dens is None:
            R[i, j] = [(np.floor(u * 10.0) + 1.0) / 10.0]                     # ratings 0.1 .. 1.0
            M[i, j] = [hash_uniform_matrix(seed + 100, n[i], n[j]) < masked]   # True = unknown
        else:
            R[i, j] = [(u < dens).astype(np.float64)]
            M[i, j] = [None]
    sim = hash_uniform_matrix(60, n['movie'], n['movie'])
    sim = -0.05 * ((sim < 0.02) | (sim.T < 0.02))
    np.fill_diagonal(sim, 0.0)
    Theta = {('user', 'user'): [lam * np.eye(n['user'])],
             ('movie', 'movie'): [lam * np.eye(n['movie']), sim]}
    return R, M, Theta, list(C5_TYPES), c


def g0_from(z, prefix, types):
    return {(t, t): z['%sG0_%s' % (prefix, t)] for t in types}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


class Snapshots:
    """callback(G, S, it) recorder mirroring the golden capture."""

    def __init__(self, keep):
        self.keep, self.snap = set(keep), {}

    def __call__(self, G, S, it):
        if it in self.keep:
            self.snap[it] = ({r: np.array(v) for r, v in G.items()},
                             {r: [np.array(s) for s in v] for r, v in S.items()})


def compare_snapshots(z, prefix, snaps, tol, rows=None):
    """Compare recorded (G,S) with every `prefix`G_*_itN / S_*_itN array in the golden file."""
    worst = 0.0
    n = 0
    for key in z.files:
        if not key.startswith(prefix):
            continue
        name = key[len(prefix):]
        if name.startswith('G_') and '_it' in name:
            t, it = name[2:].rsplit('_it', 1)
            if int(it) not in snaps:
                continue
            got = snaps[int(it)][0][t, t]
            want = z[key]
            got = got[:want.shape[0]]
        elif name.startswith('S_') and '_it' in name:
            body, it = name[2:].rsplit('_it', 1)
            if int(it) not in snaps:
                continue
            i, j, l = body.rsplit('_', 2)
            got = snaps[int(it)][1][i, j][int(l)]
            want = z[key]
        else:
            continue
        e = relerr(got, want)
        worst = max(worst, e)
        n += 1
        assert e <= tol, '%s: rel err %.3e > %.1e' % (key, e, tol)
    assert n > 0, 'no golden arrays matched prefix %r' % prefix
    return worst


def fit_row_blocks(variant, R, M, Theta, types, rank, G0, max_iter, size, dtype='f64', align=None):
    """Row-block-sharded fit with `size` simulated ranks driven in lockstep from this process
    (skfusion_amd._engine.iterate_rows_lockstep); returns [(G, S) of every simulated rank]."""
    import skfusion_amd._native as nat
    from skfusion_amd._engine import (flatten_relations, flatten_thetas, count_objects,
                                      iterate_rows_lockstep)
    from skfusion_amd.fusion.decomposition._dfmf import row_block_plan
    code = {'dfmf': nat.SKF_DFMF, 'dfmc': nat.SKF_DFMC}[variant]
    n = count_objects(types, R)
    rel = flatten_relations(R, M if variant == 'dfmc' else None)
    th = flatten_thetas(Theta)
    plans = [row_block_plan(code, rel, th, types, n, rank, dtype, None, q, size) for q in range(size)]
    try:
        for p in plans:
            for t in types:
                p.set_factor(t, G0[t, t])
        iterate_rows_lockstep(plans, max_iter)
        out = []
        for p in plans:
            G = {(t, t): p.get_factor(t) for t in types}
            S = {}
            for k, (i, j, _, _) in enumerate(rel):
                S.setdefault((i, j), []).append(p.get_backbone(k))
            out.append((G, S))
        return out
    finally:
        for p in plans:
            p.close()


class ThreadGroup(object):
    """`world` plans of ONE process as the ranks of a group (test vehicle of skf_iterate_dist where there is one device or
    none): every plan is driven by its own thread, the collective callback of each meets the others at a barrier and rank 0
    does the arithmetic on the tensor views of all workspaces.  `serial` (the host emulator is not re-entrant): one thread
    inside the library at a time -- the lock is handed over while a thread waits at the barrier."""

    def __init__(self, world, sync=None, serial=True):
        import threading
        self.world, self.sync = world, sync or (lambda: None)
        self.barrier = threading.Barrier(world)
        self.lock = threading.Lock() if serial else None
        self.views = [None] * world
        self.calls = []                           # (op, elements per rank, element size) of every collective of rank 0

    def _meet(self):
        if self.lock:
            self.lock.release()
        try:
            self.barrier.wait(timeout=600)
        finally:
            if self.lock:
                self.lock.acquire()

    def collective(self, op, view, count, rank, world):
        self.sync()
        self.views[rank] = view
        if rank == 0:
            self.calls.append((op, int(count), view.element_size()))
        self._meet()
        if rank == 0:
            vs = self.views
            if op in (0, 1):                      # (all-reduce of the whole buffer covers the reduce-scatter's range)
                total = vs[0].clone()
                for v in vs[1:]:
                    total += v
                for v in vs:
                    v.copy_(total)
            else:
                for src in range(world):
                    chunk = vs[src][src * count:(src + 1) * count].clone()
                    for v in vs:
                        v[src * count:(src + 1) * count] = chunk
            self.sync()
        self._meet()

    def run(self, plans, fn):
        """fn(plan) for every plan, each in its own thread; raises the first error."""
        import threading
        errors = []

        def drive(plan):
            if self.lock:
                self.lock.acquire()
            try:
                fn(plan)
            except BaseException as exc:          # noqa: B902  (surfaced after the join)
                errors.append(exc)
                self.barrier.abort()
            finally:
                if self.lock:
                    self.lock.release()
        threads = [threading.Thread(target=drive, args=(p,)) for p in plans]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def bytes_sent_per_rank(self):
        """Ring accounting of what ONE rank sent in the collectives seen so far (skf_exchange_bytes' convention)."""
        f = (self.world - 1) / float(self.world)
        return sum((2.0 if op == 0 else 1.0) * f * (n if op == 0 else n * self.world) * es for op, n, es in self.calls)


def fit_owned(variant, R, M, Theta, types, rank, G0, max_iter, size, dtype='f64', calls=1, group=None, sqerr=None):
    """Ownership-sharded fit (SKF_OPT_OWNED_ROWS) with `size` ranks of this process driven through skf_iterate_dist by a
    ThreadGroup; `calls` calls of max_iter / calls iterations each.  Returns [(G, S) of every rank] (+ the group)."""
    import skfusion_amd._native as nat
    from skfusion_amd._engine import flatten_relations, flatten_thetas, count_objects
    from skfusion_amd.fusion.decomposition._dfmf import owned_plan
    code = {'dfmf': nat.SKF_DFMF, 'dfmc': nat.SKF_DFMC}[variant]
    n = count_objects(types, R)
    rel = flatten_relations(R, M if variant == 'dfmc' else None)
    th = flatten_thetas(Theta)
    rt = nat.get_runtime()
    grp = group or ThreadGroup(size, sync=rt.mem.synchronize, serial=True)
    plans = [owned_plan(code, rel, th, types, n, rank, dtype, None, q, size) for q in range(size)]
    try:
        for q, p in enumerate(plans):
            p.attach_callback_comm(q, size, grp.collective)
            for t in types:
                p.set_factor(t, G0[t, t])
        for _ in range(calls):
            grp.run(plans, lambda p: p.iterate_dist(max_iter // calls))
        out = []
        for p in plans:
            G = {(t, t): p.get_factor(t) for t in types}
            S = {}
            for k, (i, j, _, _) in enumerate(rel):
                S.setdefault((i, j), []).append(p.get_backbone(k))
            out.append((G, S))
        out_bytes = [p.exchange_bytes(size) for p in plans]
        if sqerr is not None:                       # squared errors of every relation: each rank holds those of ITS rows
            sqerr.extend([[p.relation_sqerr(k) for k in range(len(rel))] for p in plans])
        return out, grp, out_bytes
    finally:
        for p in plans:
            p.close()


# ---- measured deviations -------------------------------------------------------------------------------
# `within(value, bound, what)` asserts value < bound and records (what, value, bound); conftest writes the
# records of a session to gpurun_out/test_deviations.txt, so the bounds in the GPU tests can be kept at a
# small multiple of what the hardware actually measures (and the measured figure quoted next to them).
DEVIATIONS = []


def within(value, bound, what):
    value = float(value)
    DEVIATIONS.append((what, value, float(bound)))
    assert value < bound, '%s: measured %.3e, bound %.3e' % (what, value, bound)
    return value
