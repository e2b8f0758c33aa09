"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

Container-only tool (needs /root/reference); its outputs -- plain input/output arrays --
are committed so that the oracle and the HIP engine can be checked anywhere.
    python tools/gen_golden.py            # regenerates every tests/golden/*.npz

Capture points: the functional seam `dfmf(**params)` / `dfmc` / `transform`
(reference _dfmf.py:127, _dfmc.py:181, _dfmf.py:330).  `initialize` is wrapped to record G0
(this removes the dependence on set-iteration order / PYTHONHASHSEED); the reference's own
`callback` hook records (G, S) per iteration.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(1, ROOT)
from oracle_shim import load_reference          # noqa: E402

skf, ref_dfmf, ref_dfmc = load_reference()
from oracle.dfmf_oracle import hash_uniform_matrix   # noqa: E402  (synthetic inputs only)

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


class Recorder:
    """Wraps mod.initialize to capture G0 and provides the per-iteration callback."""

    def __init__(self, mod, keep_iters, g_rows=None):
        self.mod, self.keep, self.g_rows = mod, set(keep_iters), g_rows
        self.G0, self.snap = None, {}
        self._orig = mod.initialize

    def __enter__(self):
        def init(*a, **k):
            G = self._orig(*a, **k)
            self.G0 = {r: v.copy() for r, v in G.items()}
            return G
        self.mod.initialize = init
        return self

    def __exit__(self, *exc):
        self.mod.initialize = self._orig

    def callback(self, G, S, it):
        if it in self.keep:
            rows = self.g_rows
            self.snap[it] = (
                {r: (v.copy() if rows is None else v[:rows].copy()) for r, v in G.items()},
                {r: [s.copy() for s in v] for r, v in S.items()})


def pack(prefix, rec, out, errs=None):
    for (t, _), v in rec.G0.items():
        out['%sG0_%s' % (prefix, t)] = v
    for it, (G, S) in rec.snap.items():
        for (t, _), v in G.items():
            out['%sG_%s_it%d' % (prefix, t, it)] = v
        for (i, j), lst in S.items():
            for l, s in enumerate(lst):
                out['%sS_%s_%s_%d_it%d' % (prefix, i, j, l, it)] = s
    if errs is not None:
        for (i, j), lst in errs.items():
            out['%serr_%s_%s' % (prefix, i, j)] = np.asarray(lst)


def fro_errs(R, G, S):
    return {(i, j): [np.linalg.norm(Rl - G[i, i].dot(S[i, j][l]).dot(G[j, j].T))
                     for l, Rl in enumerate(mats)] for (i, j), mats in R.items()}


def save(name, out):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **out)
    print('%-28s %8.1f KB  (%d arrays)' % (name, os.path.getsize(path) / 1024., len(out)))


# ------------------------------------------------------------------ C1: README graph
def readme_graph():
    R12 = np.random.RandomState(0).rand(50, 100)      # README.md:50-64
    R13 = np.random.RandomState(1).rand(50, 40)
    R23 = np.random.RandomState(2).rand(100, 40)
    R = {('t1', 't2'): [R12], ('t1', 't3'): [R13], ('t2', 't3'): [R23]}
    return R, ['t1', 't2', 't3'], {'t1': 10, 't2': 20, 't3': 30}


def gen_c1():
    R, types, rank = readme_graph()
    out = {}
    for init in ('random', 'random_c', 'random_vcol'):
        with Recorder(ref_dfmf, (0, 1, 9, 99)) as rec:
            G, S = ref_dfmf.dfmf(R, {}, types, rank, max_iter=100, init_type=init,
                                 callback=rec.callback,
                                 random_state=np.random.RandomState(0))
        pack(init + '/', rec, out, fro_errs(R, G, S))
    save('c1_readme_dfmf.npz', out)
    return G, S     # random_vcol run, used by the transform golden


# ------------------------------------------------------------------ multi-relation probe graph
def probe_graph():
    rs = np.random.RandomState(7)
    n1, n2, n3 = 40, 30, 20
    R = {('t1', 't2'): [rs.rand(n1, n2), rs.rand(n1, n2) * (rs.rand(n1, n2) > 0.7)],
         ('t1', 't3'): [rs.randn(n1, n3)],                         # negative-valued relation
         ('t2', 't3'): [rs.rand(n2, n3)]}
    th1 = np.zeros((n1, n1))
    idx = rs.randint(0, n1, size=(60, 2))
    th1[idx[:, 0], idx[:, 1]] = rs.uniform(-0.1, 0.05, size=60)
    th1 = 0.5 * (th1 + th1.T)
    th2a = -0.05 * (rs.rand(n2, n2) > 0.9)
    th2b = 0.02 * np.eye(n2)
    Theta = {('t1', 't1'): [th1], ('t2', 't2'): [th2a, th2b]}
    M = {('t1', 't2'): [rs.rand(n1, n2) > 0.8, None],
         ('t1', 't3'): [None],
         ('t2', 't3'): [rs.rand(n2, n3) > 0.5]}
    return R, Theta, M, ['t1', 't2', 't3'], {'t1': 6, 't2': 5, 't3': 4}


def gen_probe():
    R, Theta, M, types, rank = probe_graph()
    out = {}
    for (i, j), mats in R.items():
        for l, m in enumerate(mats):
            out['R_%s_%s_%d' % (i, j, l)] = m
            if M[i, j][l] is not None:
                out['M_%s_%s_%d' % (i, j, l)] = M[i, j][l]
    for (i, _), ths in Theta.items():
        for l, th in enumerate(ths):
            out['Theta_%s_%d' % (i, l)] = th
    with Recorder(ref_dfmf, (0, 1, 9, 29)) as rec:
        G, S = ref_dfmf.dfmf(R, Theta, types, rank, max_iter=30, init_type='random_vcol',
                             callback=rec.callback, random_state=np.random.RandomState(3))
    pack('dfmf/', rec, out, fro_errs(R, G, S))
    Rin = {k: [m.copy() for m in v] for k, v in R.items()}
    with Recorder(ref_dfmc, (0, 1, 9, 29)) as rec:
        G, S = ref_dfmc.dfmc(Rin, M, Theta, types, rank, max_iter=30, init_type='random_vcol',
                             callback=rec.callback, random_state=np.random.RandomState(3))
    for k in R:        # tests/test_dfmc.py:62,85 -- inputs must not be mutated
        for a, b in zip(R[k], Rin[k]):
            assert np.array_equal(a, b)
    pack('dfmc/', rec, out, fro_errs(R, G, S))
    save('probe_multirel.npz', out)


# ------------------------------------------------------------------ rank-deficient Gram
def gen_rank_deficient():
    rs = np.random.RandomState(0)                      # tests/test_n_run.py:10-16
    R12, R13 = rs.rand(30, 30), rs.rand(30, 30)
    R = {('t1', 't2'): [R12], ('t1', 't3'): [R13]}
    types, rank = ['t1', 't2', 't3'], {'t1': 50, 't2': 30, 't3': 10}
    out = {'R_t1_t2_0': R12, 'R_t1_t3_0': R13}
    with Recorder(ref_dfmf, (0, 1, 9, 99)) as rec:
        G, S = ref_dfmf.dfmf(R, {}, types, rank, max_iter=100, init_type='random',
                             callback=rec.callback, random_state=np.random.RandomState(5))
    pack('dfmf/', rec, out, fro_errs(R, G, S))
    M = {k: [None] for k in R}
    with Recorder(ref_dfmc, (0, 1, 9, 99)) as rec:
        G, S = ref_dfmc.dfmc(R, M, {}, types, rank, max_iter=100, init_type='random',
                             callback=rec.callback, random_state=np.random.RandomState(5))
    pack('dfmc/', rec, out, fro_errs(R, G, S))
    save('rank_deficient.npz', out)


# ------------------------------------------------------------------ transform (fold-in)
def gen_transform(G, S):
    rs = np.random.RandomState(11)                     # README.md:75-87 shapes
    new12, new13 = rs.rand(10, 100), rs.rand(10, 40)
    new21 = rs.rand(100, 10)                           # target as COLUMN type
    th = -0.05 * (rs.rand(10, 10) > 0.7)
    t1 = 't1'
    rank = {'t1': 10, 't2': 20, 't3': 30}
    out = {'new_t1_t2': new12, 'new_t1_t3': new13, 'new_t2_t1': new21, 'theta_t1': th}
    for (t, _), v in G.items():
        out['G_%s' % t] = v
    for (i, j), lst in S.items():
        out['S_%s_%s' % (i, j)] = lst[0]
    # S for the reversed pair (t2,t1): use S_12^T so that the column branch is exercised
    S2 = dict(S)
    S2[('t2', t1)] = [S[(t1, 't2')][0].T.copy()]
    out['S_t2_t1'] = S2[('t2', t1)][0]
    Rn = {(t1, 't2'): [new12], (t1, 't3'): [new13], ('t2', t1): [new21]}
    for init in ('random_c', 'random_vcol', 'random'):
        snaps = {}
        g0 = {}
        orig = ref_dfmf.initialize

        def init_wrap(*a, **k):
            Gx = orig(*a, **k)
            g0['G0'] = Gx[t1, t1].copy()
            return Gx
        ref_dfmf.initialize = init_wrap
        try:
            Gi = ref_dfmf.transform(Rn, {(t1, t1): [th]}, t1, rank, G, S2, max_iter=100,
                                    init_type=init, random_state=np.random.RandomState(4),
                                    callback=lambda g, it: snaps.__setitem__(it, g.copy()))
        finally:
            ref_dfmf.initialize = orig
        out[init + '/G0'] = g0['G0']
        for it in (0, 9, 99):
            out['%s/G_it%d' % (init, it)] = snaps[it]
        assert np.array_equal(Gi, snaps[99])
    save('transform_readme.npz', out)


# ------------------------------------------------------------------ C2: dicty
def gen_dicty():
    g = skf.datasets.load_dicty()
    ann, expr, ppi = g['ann'], g['expr'], g['ppi']
    gene, go, exc = ann.row_type, ann.col_type, expr.col_type
    # inputs as data fixture (values only; ann is 0/1, ppi is 3% dense)
    pr, pc = np.nonzero(ppi.data)
    # expression values have <= 3 decimals in the csv: keep them exactly as integer milli-units;
    # the loader re-applies log(max(x, eps)) (reference datasets/base.py:57)
    raw = np.exp(expr.data)
    milli = np.rint(raw * 1000.0).astype(np.int64)
    x = milli / 1000.0
    assert np.array_equal(np.log(np.maximum(x, np.finfo(float).eps)), expr.data)
    inp = {'ann_bits': np.packbits(ann.data.astype(bool), axis=1), 'ann_shape': np.array(ann.data.shape),
           'expr_milli': milli.astype(np.int32), 'ppi_rows': pr.astype(np.int32), 'ppi_cols': pc.astype(np.int32),
           'ppi_vals': ppi.data[pr, pc], 'ppi_shape': np.array(ppi.data.shape),
           'ranks': np.array([gene.rank, go.rank, exc.rank])}
    assert set(np.unique(ann.data)) <= {0.0, 1.0}
    save('dicty_inputs.npz', inp)

    types, rank = ['gene', 'go', 'exc'], {'gene': 50, 'go': 15, 'exc': 5}
    R = {('gene', 'go'): [ann.data], ('gene', 'exc'): [expr.data]}
    Theta = {('gene', 'gene'): [ppi.data]}
    out = {}
    with Recorder(ref_dfmf, (0, 9, 99)) as rec:
        G, S = ref_dfmf.dfmf(R, Theta, types, rank, max_iter=100, init_type='random_vcol',
                             callback=rec.callback, random_state=np.random.RandomState(0))
    # keep the file small: full G only for G0 and the final iterate
    for it in (0, 9):
        Gs, Ss = rec.snap[it]
        rec.snap[it] = ({r: v[:32].copy() for r, v in Gs.items()}, Ss)
    pack('dfmf/', rec, out, fro_errs(R, G, S))
    # Dfmc with a row-block mask on `ann` (examples/dicty_association.py:37-44 pattern)
    mask = np.zeros(ann.data.shape, dtype=bool)
    mask[:200, :] = True
    out['dfmc/mask_rows'] = np.array([0, 200])
    M = {('gene', 'go'): [mask], ('gene', 'exc'): [None]}
    with Recorder(ref_dfmc, (0, 9, 29), g_rows=32) as rec:
        G, S = ref_dfmc.dfmc(R, M, Theta, types, rank, max_iter=30, init_type='random_vcol',
                             callback=rec.callback, random_state=np.random.RandomState(0))
    g0_same = all(np.array_equal(out['dfmf/G0_%s' % t], rec.G0[t, t]) for t in types)
    assert g0_same       # same seed, same init -> reuse dfmf/G0 in the tests
    rec.G0 = {}
    pack('dfmc/', rec, out, None)
    out['dfmc/G_gene_final_rows'] = G['gene', 'gene'][:256].copy()
    save('c2_dicty.npz', out)


# ------------------------------------------------------------------ C3 scaled 1/25
def gen_c3_scaled():
    n1, n2, n3 = 2000, 4000, 1600
    R12 = hash_uniform_matrix(0, n1, n2)
    R13 = hash_uniform_matrix(1, n1, n3)
    R23 = hash_uniform_matrix(2, n2, n3)
    rank = {'t1': 128, 't2': 256, 't3': 256}
    types = ['t1', 't2', 't3']
    G0 = {('t1', 't1'): hash_uniform_matrix(100, n1, 128),
          ('t2', 't2'): hash_uniform_matrix(101, n2, 256),
          ('t3', 't3'): hash_uniform_matrix(102, n3, 256)}
    R = {('t1', 't2'): [R12], ('t1', 't3'): [R13], ('t2', 't3'): [R23]}
    orig = ref_dfmf.initialize
    ref_dfmf.initialize = lambda *a, **k: {r: v.copy() for r, v in G0.items()}
    errs_it = {}
    snaps = {}

    def cb(G, S, it):
        e = fro_errs(R, G, S)
        errs_it[it] = [e[k][0] for k in sorted(e)]
        snaps[it] = ({r: v[:16].copy() for r, v in G.items()},
                     {r: [s.copy() for s in v] for r, v in S.items()} if it == 4 else {})
    try:
        G, S = ref_dfmf.dfmf(R, {}, types, rank, max_iter=5, init_type='random',
                             callback=cb, random_state=np.random.RandomState(0))
    finally:
        ref_dfmf.initialize = orig
    out = {'shape': np.array([n1, n2, n3]), 'ranks': np.array([128, 256, 256]),
           'data_seeds': np.array([0, 1, 2]), 'g0_seeds': np.array([100, 101, 102]),
           'errs': np.array([errs_it[i] for i in range(5)])}
    for it, (Gs, Ss) in snaps.items():
        for (t, _), v in Gs.items():
            out['Grows_%s_it%d' % (t, it)] = v
        for (i, j), lst in Ss.items():
            out['S_%s_%s_it%d' % (i, j, it)] = lst[0]
    save('c3_scaled.npz', out)


# ------------------------------------------------------------------ C3 planted, 1/25 scale
def gen_planted():
    """The reference on the planted variant of config 3 at 1/25 linear scale (tests/helpers.py:c3_planted_graph; inputs
    regenerable from seeds), 60 iterations from the hash-generated G0: per-relation Frobenius errors at iterations 10, 30,
    60, the backbones and 16 rows of every factor at iteration 60 -- once on the fp64 relations and once on their bf16
    roundings (what the SKF_BF16 engine stores), plus the quantisation term ||bf16(R) - R||_F of every relation."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import c3_planted_graph
    out = {}
    keep = (9, 29, 59)
    for tag, bf16 in (('f64', False), ('bf16', True)):
        R, G0, types, rank = c3_planted_graph(bf16=bf16)
        if not bf16:
            Rexact = R
            out['shape'] = np.array([R['t1', 't2'][0].shape[0], R['t1', 't2'][0].shape[1], R['t1', 't3'][0].shape[1]])
            out['ranks'] = np.array([rank[t] for t in types])
        else:
            out['quantisation'] = np.array([np.linalg.norm(R[k][0] - Rexact[k][0]) for k in sorted(R)])
        orig = ref_dfmf.initialize
        ref_dfmf.initialize = lambda *a, **k: {r: v.copy() for r, v in G0.items()}
        errs_it, last = {}, {}

        def cb(G, S, it):
            if it in keep:
                e = fro_errs(R, G, S)
                errs_it[it] = [e[k][0] for k in sorted(e)]
            if it == keep[-1]:
                last['G'] = {r: v[:16].copy() for r, v in G.items()}
                last['S'] = {r: [s.copy() for s in v] for r, v in S.items()}
        try:
            ref_dfmf.dfmf(R, {}, types, rank, max_iter=keep[-1] + 1, init_type='random', callback=cb,
                          random_state=np.random.RandomState(0))
        finally:
            ref_dfmf.initialize = orig
        out['%s/iters' % tag] = np.array(keep)
        out['%s/errs' % tag] = np.array([errs_it[i] for i in keep])
        if not bf16:
            for (t, _), v in last['G'].items():
                out['f64/Grows_%s' % t] = v
            for (i, j), lst in last['S'].items():
                out['f64/S_%s_%s' % (i, j)] = lst[0]
    save('c3_planted_scaled.npz', out)


# ------------------------------------------------------------------ C5: MovieLens-style Dfmc
def gen_c5():
    """BASELINE config 5 scaled down (tests/helpers.py:movielens_style_graph): inputs are regenerated
    from seeds, so only G0 and the reference's (G, S) snapshots / errors are stored."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import movielens_style_graph
    R, M, Theta, types, rank = movielens_style_graph()
    Rin = {k: [m.copy() for m in v] for k, v in R.items()}
    out = {}
    with Recorder(ref_dfmc, (0, 1, 9, 29)) as rec:
        G, S = ref_dfmc.dfmc(Rin, M, Theta, types, rank, max_iter=30, init_type='random_vcol',
                             callback=rec.callback, random_state=np.random.RandomState(5))
    for k in R:
        for a, b in zip(R[k], Rin[k]):
            assert np.array_equal(a, b)
    pack('dfmc/', rec, out, fro_errs(R, G, S))
    known = ~M['user', 'movie'][0]
    rec_um = G['user', 'user'].dot(S['user', 'movie'][0]).dot(G['movie', 'movie'].T)
    out['dfmc/rmse_known'] = np.sqrt(np.mean((rec_um - R['user', 'movie'][0])[known] ** 2))
    out['dfmc/rmse_unknown'] = np.sqrt(np.mean((rec_um - R['user', 'movie'][0])[~known] ** 2))
    save('c5_movielens_scaled.npz', out)


# ------------------------------------------------------------------ fill strategies (host glue)
def gen_fill():
    """Relation.filled() of the reference (fusion_graph.py:464-545) on inputs with NaN / inf /
    masked entries: data and mask of the result for the four strategies."""
    from skfusion.fusion import Relation, ObjectType
    rs = np.random.RandomState(0)
    x = rs.rand(5, 4)
    xm = np.ma.masked_greater(x.copy(), 0.7)
    xm[xm < 0.2] = np.nan
    xm[1, 1] = np.inf
    y = x.copy()
    y[y < 0.2] = np.nan
    y[3, 2] = -np.inf
    z = np.ma.masked_greater(rs.rand(6, 5), 0.6)            # masked, all finite
    # corners the three inputs above do not reach: NaN / inf UNDER the mask, a row that is unknown throughout (masked, NaN),
    # a column unknown throughout, +inf and -inf in one row
    w = rs.rand(7, 6)
    wm = np.ma.masked_greater(w.copy(), 0.75)
    wm.data[np.ma.getmaskarray(wm) & (w > 0.9)] = np.nan        # non-finite values beneath the mask
    wm.data[0, 0] = np.inf
    wm[0, 0] = np.ma.masked
    wm[2, :] = np.ma.masked                                     # row 2: masked throughout
    wm.data[4, :] = np.nan                                      # row 4: NaN throughout, where not masked
    wm.data[:, 3] = np.nan                                      # column 3 likewise
    wm.data[5, 1], wm.data[5, 2] = np.inf, -np.inf
    wm.mask[5, 1] = wm.mask[5, 2] = False
    v = rs.rand(6, 5)                                           # plain: a NaN row, a NaN column, both infinities
    v[1, :] = np.nan
    v[:, 4] = np.nan
    v[3, 0], v[3, 1] = np.inf, -np.inf
    # round 6 (advisor): a MaskedArray that masks NOTHING (numpy.ma.is_masked false) with infinite / NaN line means --
    # numpy.ma masks an infinite mean itself and the reference then leaves those entries masked with 1 beneath -- and a
    # genuinely masked input with infinities in its lines
    u = rs.rand(6, 5)
    u[0, 1] = np.inf                                            # row 0 / column 1: an infinite mean
    u[2, 0], u[2, 3] = np.inf, -np.inf                          # row 2: both infinities -> NaN mean -> matrix mean
    u[4, :] = np.nan                                            # row 4 without entries
    u[1, 2] = u[3, 4] = np.nan
    um = np.ma.MaskedArray(u.copy(), mask=np.zeros(u.shape, dtype=bool))
    u2 = rs.rand(4, 5)                                          # the same shape with a FINITE matrix mean: one NaN row
    u2[1, :] = np.nan
    u2[0, 2] = u2[3, 3] = np.nan
    um2 = np.ma.MaskedArray(u2.copy(), mask=np.zeros(u2.shape, dtype=bool))
    q = rs.rand(6, 5)
    qm = np.ma.masked_greater(q.copy(), 0.7)
    qm.data[1, 1] = np.inf
    qm.mask[1, 1] = False
    qm.data[3, 0], qm.data[3, 2] = -np.inf, np.nan
    qm.mask[3, 0] = qm.mask[3, 2] = False
    out = {'masked_data': xm.data, 'masked_mask': np.ma.getmaskarray(xm), 'plain': y,
           'finite_data': z.data, 'finite_mask': np.ma.getmaskarray(z),
           'corner_data': wm.data.copy(), 'corner_mask': np.ma.getmaskarray(wm).copy(), 'plaincorner': v,
           'nomask_data': um.data.copy(), 'nomask_mask': np.ma.getmaskarray(um).copy(),
           'nomaskfinite_data': um2.data.copy(), 'nomaskfinite_mask': np.ma.getmaskarray(um2).copy(),
           'maskedinf_data': qm.data.copy(), 'maskedinf_mask': np.ma.getmaskarray(qm).copy()}
    t1, t2 = ObjectType('a'), ObjectType('b')
    import warnings
    for tag, arr in (('masked', xm), ('plain', y), ('finite', z), ('corner', wm), ('plaincorner', v), ('nomask', um),
                     ('nomaskfinite', um2), ('maskedinf', qm)):
        for fv in ('mean', 'row_mean', 'col_mean', 0.5):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                f = Relation(arr.copy(), t1, t2, fill_value=fv).filled()
            key = '%s/%s' % (tag, fv)
            out[key + '/data'] = np.ma.getdata(f)
            out[key + '/mask'] = np.ma.getmaskarray(f)
            out[key + '/is_masked'] = np.array(bool(np.ma.is_masked(f)))
    save('fill_strategies.npz', out)


# ------------------------------------------------------------------ f4: chained latent profiles
def chain_graph_data():
    """Four types with multi-hop paths from A: A->B, A->C, B->C, B->D, C->D (+ a constraint on A), new A objects for the
    fold-in.  (sizes, ranks, matrices) from one seed -- tests/helpers.chain_graph rebuilds the same arrays."""
    rs = np.random.RandomState(21)
    sizes = {'A': 60, 'B': 45, 'C': 30, 'D': 25}
    ranks = {'A': 7, 'B': 5, 'C': 6, 'D': 4}
    mats = {('A', 'B'): rs.rand(60, 45), ('A', 'C'): rs.rand(60, 30), ('B', 'C'): rs.rand(45, 30),
            ('B', 'D'): rs.rand(45, 25), ('C', 'D'): rs.rand(30, 25)}
    th = np.zeros((60, 60))
    idx = rs.randint(0, 60, size=(40, 2))
    th[idx[:, 0], idx[:, 1]] = -0.05
    th = 0.5 * (th + th.T)
    new = {('A', 'B'): rs.rand(12, 45), ('A', 'C'): rs.rand(12, 30)}
    return sizes, ranks, mats, th, new


def gen_chain():
    """The REFERENCE classes fit the graph and fold new objects in; the profiles are computed with the arithmetic of the
    reference's own examples (examples/dicty_chaining.py:40-53 `profile`: G_row . reduce(dot, backbones) . G_col^T, one type
    skipped; examples/pharma_chaining.py:43-53: G_row . reduce(dot, backbones)) on the reference's (G, S), for the fit's own
    objects and for the transformer's new ones.  Stored: every factor, every backbone, the transformer's factor, the
    path list (type names) and the four profile matrices."""
    from functools import reduce
    sizes, ranks, mats, th, new = chain_graph_data()
    f = skf.fusion
    ot = {k: f.ObjectType(k, ranks[k]) for k in 'ABCD'}
    rels = {k: f.Relation(m, ot[k[0]], ot[k[1]]) for k, m in mats.items()}
    graph = f.FusionGraph(list(rels.values()) + [f.Relation(th, ot['A'], ot['A'])])
    fuser = f.Dfmf(max_iter=30, init_type='random', random_state=np.random.RandomState(3)).fuse(graph)
    tgraph = f.FusionGraph([f.Relation(new[k], ot[k[0]], ot[k[1]]) for k in new])
    transformer = f.DfmfTransform(max_iter=30, init_type='random', random_state=np.random.RandomState(4))
    transformer.transform(ot['A'], tgraph, fuser)
    order = [ot[k] for k in 'ABCD']

    def profile(fuser, transformer, project, skip=None):
        X, paths = [], []
        for obj_type in order:
            for c in fuser.chain(ot['A'], obj_type):
                if obj_type is skip:
                    continue
                cf = [fuser.backbone(fuser.fusion_graph[c[i]][c[i + 1]][0]) for i in range(len(c) - 1)]
                bb = reduce(np.dot, cf) if cf != [] else []
                row_factor = transformer.factor(ot['A'])
                if project:
                    obj_factor = fuser.factor(obj_type)
                    X.append(np.dot(row_factor, np.dot(bb, obj_factor.T)) if len(cf) else row_factor)
                else:
                    X.append(np.dot(row_factor, bb) if len(cf) else row_factor)
                paths.append('>'.join(t.name for t in c))
        return np.hstack(X), paths

    out = {}
    for k in 'ABCD':
        out['G_%s' % k] = fuser.factor(ot[k])
    for k, r in rels.items():
        out['S_%s_%s' % k] = fuser.backbone(r)
    out['G_new_A'] = transformer.factor(ot['A'])
    X, paths = profile(fuser, fuser, True, skip=ot['B'])
    out['profile/fit_project_skipB'], out['paths/skipB'] = X, np.array(paths)
    X, paths = profile(fuser, fuser, False)
    out['profile/fit_plain'], out['paths/all'] = X, np.array(paths)
    out['profile/new_project_skipB'] = profile(fuser, transformer, True, skip=ot['B'])[0]
    out['profile/new_plain'] = profile(fuser, transformer, False)[0]
    assert len(out['paths/all']) == 7 and 'A>B>C>D' in set(out['paths/all'])
    save('chain_profiles.npz', out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['c1', 'probe', 'rd', 'transform', 'dicty', 'c3s', 'c3p', 'c5', 'fill', 'chain']
    G = S = None
    if 'c1' in which or 'transform' in which:
        G, S = gen_c1()
    if 'probe' in which:
        gen_probe()
    if 'rd' in which:
        gen_rank_deficient()
    if 'transform' in which:
        gen_transform(G, S)
    if 'dicty' in which:
        gen_dicty()
    if 'c3s' in which:
        gen_c3_scaled()
    if 'c3p' in which:
        gen_planted()
    if 'c5' in which:
        gen_c5()
    if 'fill' in which:
        gen_fill()
    if 'chain' in which:
        gen_chain()
