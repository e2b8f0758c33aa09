"""Functional solver seam of DFMF and of the fold-in transform -- same signatures as the
reference's ``dfmf()`` (_dfmf.py:127-129) and ``transform()`` (_dfmf.py:330-333) -- driving the
device engine.  ``R`` / ``Theta`` are dictionaries keyed by (row_type, col_type) whose values
are LISTS of matrices; the result is ``(G, S)`` with ``G[(t, t)]`` and ``S[(i, j)] = [..]``.

Extra keyword-only engine options (defaults keep the reference behaviour):
  dtype   'f64' (default, parity with the float64 reference) | 'f32'
  G0      dict {(t,t): ndarray} overriding the initialiser (warm start / parity tests)
"""
import logging

import numpy as np

from ... import _native as nat
from ..._engine import DevicePlan, flatten_relations, flatten_thetas, count_objects
from ._init import initialize

log = logging.getLogger('skfusion_amd')


def _as_rs(random_state):
    if isinstance(random_state, np.random.RandomState):
        return random_state
    return np.random.RandomState(random_state)


def _unpack(values, obj_types, rel_list):
    G = {(t, t): values[k] for k, t in enumerate(obj_types)}
    S = {}
    for k, rel in enumerate(rel_list):
        S.setdefault((rel[0], rel[1]), []).append(values[len(obj_types) + k])
    return G, S


def _collect(plan, obj_types, rel_list):
    """(G, S) of the plan: all copies issued, ONE synchronisation, then the read-back."""
    fetched = plan.fetch_results(obj_types, len(rel_list))
    plan.synchronize()
    return _unpack(plan.read_results(fetched), obj_types, rel_list)


def _rel_index(rel_list, target):
    """target = (row, col) or ((row, col), l) -> flat relation index."""
    if isinstance(target[0], tuple):
        pair, l = target
    else:
        pair, l = target, 0
    hits = [k for k, (i, j, _, _) in enumerate(rel_list) if (i, j) == tuple(pair)]
    return hits[l]


def _host_loop(step, sqerrs, rel_list, max_iter, stopping, stopping_system, compute_err, on_iter):
    """The body of the reference loops with their early-stopping / error logging (_dfmf.py:212-221, 301-322;
    _dfmc.py:270-279, 370-392; fold-in _dfmf.py:367-376, 433-450), one device iteration per pass.
    step(): one iteration; sqerrs(indices): squared reconstruction errors of those relations (summed over the
    ranks of a sharded fit, so that every rank takes the same decision); on_iter(it): the callback hook."""
    if stopping_system:
        compute_err = True
    err_t = (None, None)
    err_s = (None, None)
    objective = []
    for it in range(max_iter):
        if it > 1 and stopping and err_t[1] - err_t[0] < stopping[1]:
            log.info("Early stopping: target matrix change < %5.4f", stopping[1])
            break
        if it > 1 and stopping_system and err_s[1] - err_s[0] < stopping_system:
            log.info("Early stopping: matrix system change < %5.4f", stopping_system)
            break
        step()
        if compute_err:
            sq = sqerrs(list(range(len(rel_list))))
            s = 0.
            for (i, j, _, _), v in zip(rel_list, sq):
                e = np.sqrt(v)
                log.info("Relation R_%s,%s norm difference: %5.4f", i, j, e)
                s += e
            log.info("Error (objective function value): %5.4f", s)
            objective.append(s)
            err_s = (s, err_s[0])
            if stopping:
                err_t = (np.sqrt(sq[_rel_index(rel_list, stopping[0])]), err_t[0])
        elif stopping:
            k = _rel_index(rel_list, stopping[0])
            err_t = (np.sqrt(sqerrs([k])[0]), err_t[0])
        if on_iter:
            on_iter(it)
    if compute_err:
        log.info("Violations of optimization objective: %d/%d",
                 int(np.sum(np.diff(objective) > 0)), len(objective))
    return objective


def row_block_plan(variant, rel_list, theta_list, obj_types, n_obj, obj_type2rank, dtype, engine, rank, size):
    """The plan of rank `rank` of `size` in a row-block-sharded fit: all relations listed, each with
    this rank's row block of `_distributed.partition_rows` (or marked absent)."""
    from ..._distributed import partition_rows
    # block boundaries: multiples of the contraction tile for big graphs; bf16 needs multiples of 64
    align = 256 if min(n_obj.values()) >= 4096 else (64 if dtype == 'bf16' else 1)
    blocks, theta_owner = partition_rows(rel_list, theta_list, n_obj, obj_type2rank, align=align, size=size)
    from ..._engine import is_binary_matrix
    local = []
    for (i, j, m, mask), blk in zip(rel_list, blocks):
        mine = [b for b in blk if b[0] == rank]
        info = {'masked': mask is not None, 'col_side': bool(mine) and mine[0][1] == 0}
        if dtype == 'bf16' and mask is None:         # decided on the whole relation: the same path on every rank
            info['binary'] = is_binary_matrix(m)
        if not mine:
            info.update(absent=True, row_begin=0, n_rows=0, col_side=False)
            local.append((i, j, None, None, info))
            continue
        _, a, cnt = mine[0]
        info.update(absent=False, row_begin=a, n_rows=cnt)
        local.append((i, j, np.asarray(m)[a:a + cnt], None if mask is None else np.asarray(mask)[a:a + cnt], info))
    return DevicePlan(obj_types, n_obj, obj_type2rank, local,
                      [t for t, o in zip(theta_list, theta_owner) if o == rank], variant,
                      dtype=dtype, engine=engine, part=(rank, size))


def owned_plan(variant, rel_list, theta_list, obj_types, n_obj, obj_type2rank, dtype, engine, rank, size):
    """The plan of rank `rank` of `size` in a fit sharded by OWNERSHIP (SKF_OPT_OWNED_ROWS): the rank owns the same share
    of the rows of every object type (`_engine.owned_rows`) and holds exactly those rows of every relation whose row type
    it is -- work is 1 / size of every relation, the row-side terms never leave the rank -- and every constraint."""
    from ..._engine import is_binary_matrix, owned_rows, known_lists_pay
    from ..._distributed import same_on_all_ranks
    local = []
    for (i, j, m, mask) in rel_list:
        begin, count, _ = owned_rows(dtype, n_obj[i], rank, size)
        info = {'masked': mask is not None, 'col_side': True, 'row_begin': begin if count else 0, 'n_rows': count,
                'absent': count == 0}
        if mask is not None and variant == nat.SKF_DFMC:      # lists of the known entries: one decision for all ranks
            mk = np.asarray(mask, dtype=bool)
            info['known_lists'] = same_on_all_ranks(known_lists_pay(
                mk.size - int(np.count_nonzero(mk)), mk.shape[0], mk.shape[1], int(obj_type2rank[i]), dtype))
        if dtype == 'bf16' and mask is None:         # decided on the whole relation: the same path on every rank
            info['binary'] = is_binary_matrix(m)
        if count == 0:
            local.append((i, j, None, None, info))
            continue
        local.append((i, j, np.asarray(m)[begin:begin + count],
                      None if mask is None else np.asarray(mask)[begin:begin + count], info))
    return DevicePlan(obj_types, n_obj, obj_type2rank, local, list(theta_list), variant, dtype=dtype, engine=engine,
                      part=(rank, size), owned=True)


def run_fit_owned(variant, R, M, Theta, obj_types, obj_type2rank, max_iter, init_type,
                  random_state, dtype, G0, engine, stopping=None, stopping_system=None, compute_err=False,
                  callback=None):
    """One fit sharded by ownership over the ranks of the process group (`shard='owned'`; SURVEY.md 8e, second row, in
    the form with the least exchange): per iteration a rank sends the partial Q of every relation (reduce-scatter), its
    updated factor rows (all-gather) and c x c partial sums -- no E / D exchange (DevicePlan.iterate_dist, the library
    issues the collectives).  Every rank ends with the full (G, S)."""
    from ..._distributed import world, sum_over_ranks
    obj_types = list(obj_types)
    n_obj = count_objects(obj_types, R)
    if G0 is None:
        R_first = {k: np.asarray(v[0], dtype=float) for k, v in R.items()}
        G0 = initialize(obj_types, n_obj, obj_type2rank, R_first, init_type, _as_rs(random_state))
    rel_list = flatten_relations(R, M)
    rank, size = world()
    plan = owned_plan(variant, rel_list, flatten_thetas(Theta), obj_types, n_obj, obj_type2rank, dtype, engine, rank, size)
    try:
        if not plan.attach_comm():
            # a single process: every row is owned here and nothing is exchanged -- a genuine communicator of one rank (the
            # null communicator of bench.py --emulate-rank never updates the factors: not for fits)
            plan.attach_single_comm()
        for t in obj_types:
            plan.set_factor(t, G0[t, t])
        if not (callback or stopping or stopping_system or compute_err):
            plan.iterate_dist(max_iter)
        else:
            def sqerrs(idx):                   # every rank holds the squared error of ITS rows
                return sum_over_ranks([plan.relation_sqerr(k) for k in idx])
            _host_loop(lambda: plan.iterate_dist(1), sqerrs, rel_list, max_iter, stopping, stopping_system,
                       compute_err,
                       (lambda it: callback(*(_collect(plan, obj_types, rel_list) + (it,)))) if callback else None)
        return _collect(plan, obj_types, rel_list)
    finally:
        plan.close()


def run_fit_rows(variant, R, M, Theta, obj_types, obj_type2rank, max_iter, init_type,
                 random_state, dtype, G0, engine, stopping=None, stopping_system=None, compute_err=False,
                 callback=None):
    """One fit whose relations are cut into balanced ROW BLOCKS over the ranks of the process group
    (SURVEY.md 8e): every rank lists all relations with its own row block (or none), factors are
    replicated, and an iteration is four stages with all-reduces of W / Q, E / D in between
    (`DevicePlan.iterate_rows`).  Every rank ends with the full (G, S)."""
    from ..._distributed import world
    obj_types = list(obj_types)
    n_obj = count_objects(obj_types, R)
    if G0 is None:
        R_first = {k: np.asarray(v[0], dtype=float) for k, v in R.items()}
        G0 = initialize(obj_types, n_obj, obj_type2rank, R_first, init_type, _as_rs(random_state))
    rel_list = flatten_relations(R, M)
    rank, size = world()
    plan = row_block_plan(variant, rel_list, flatten_thetas(Theta), obj_types, n_obj, obj_type2rank, dtype,
                          engine, rank, size)
    try:
        plan.attach_comm()                 # the library issues the exchanges itself (skf_iterate_dist)
        for t in obj_types:
            plan.set_factor(t, G0[t, t])
        if not (callback or stopping or stopping_system or compute_err):
            plan.iterate_rows(max_iter)
        else:
            from ..._distributed import sum_over_ranks
            # every rank holds the squared error of ITS row blocks; factors and backbones are replicated

            def sqerrs(idx):
                return sum_over_ranks([plan.relation_sqerr(k) for k in idx])
            _host_loop(lambda: plan.iterate_rows(1), sqerrs, rel_list, max_iter, stopping, stopping_system,
                       compute_err,
                       (lambda it: callback(*(_collect(plan, obj_types, rel_list) + (it,)))) if callback else None)
        return _collect(plan, obj_types, rel_list)
    finally:
        plan.close()


def run_fit_sharded(variant, R, M, Theta, obj_types, obj_type2rank, max_iter, init_type,
                    random_state, dtype, G0, engine, stopping=None, stopping_system=None, compute_err=False,
                    callback=None):
    """One fit whose relations are partitioned over the ranks of the process group (one GPU
    each): replicated factors, local contractions, ONE all-reduce of the E / D accumulators per
    iteration (second row).  Every rank returns the full (G, S).
    `callback`, `stopping`, `stopping_system` and `compute_err` must be given identically on EVERY rank (as with any
    SPMD launch of the same script): the callback path gathers the backbones collectively, and a callback on one rank
    only would leave the others out of that collective.  The current device of each rank must be set before the
    fit (torch.cuda.set_device(LOCAL_RANK)) for the RCCL reductions."""
    from ..._distributed import partition_relations, gather_backbones, world
    obj_types = list(obj_types)
    n_obj = count_objects(obj_types, R)
    if G0 is None:
        R_first = {k: np.asarray(v[0], dtype=float) for k, v in R.items()}
        G0 = initialize(obj_types, n_obj, obj_type2rank, R_first, init_type, _as_rs(random_state))
    rel_list = flatten_relations(R, M)
    theta_list = flatten_thetas(Theta)
    rank, _ = world()
    rel_owner, theta_owner = partition_relations(rel_list, theta_list, n_obj, obj_type2rank)
    mine = [k for k, o in enumerate(rel_owner) if o == rank]
    plan = DevicePlan(obj_types, n_obj, obj_type2rank, [rel_list[k] for k in mine],
                      [t for t, o in zip(theta_list, theta_owner) if o == rank], variant,
                      dtype=dtype, engine=engine)
    try:
        plan.attach_comm()                 # the library issues the exchanges itself (skf_iterate_dist)
        for t in obj_types:
            plan.set_factor(t, G0[t, t])

        def collect():
            G = {(t, t): plan.get_factor(t) for t in obj_types}
            backbones = gather_backbones({k: plan.get_backbone(q) for q, k in enumerate(mine)}, len(rel_list))
            S = {}
            for k, (i, j, _, _) in enumerate(rel_list):
                S.setdefault((i, j), []).append(backbones[k])
            return G, S
        if not (callback or stopping or stopping_system or compute_err):
            plan.iterate_sharded(max_iter)
        else:
            from ..._distributed import sum_over_ranks
            where = {k: q for q, k in enumerate(mine)}          # global relation index -> index in this rank's plan

            def sqerrs(idx):                                     # the owner of a relation contributes its error
                return sum_over_ranks([plan.relation_sqerr(where[k]) if k in where else 0.0 for k in idx])
            _host_loop(lambda: plan.iterate_sharded(1), sqerrs, rel_list, max_iter, stopping, stopping_system,
                       compute_err, (lambda it: callback(*(collect() + (it,)))) if callback else None)
        return collect()
    finally:
        plan.close()


def run_fits_concurrent(variant, R, M, Theta, obj_types, obj_type2rank, max_iter, dtype, G0_list, engine,
                        n_streams, batch_only=False):
    """The restarts of ONE graph on one GPU, `n_streams` of them at a time -- the reference's `n_jobs` over `n_run`
    (joblib workers, dfmf.py:87-95) for graphs too small to fill the chip.  Where the plans run the schedule for small
    graphs the restarts of a batch share every launch (skf_iterate_batch: the restart is a grid dimension); otherwise each
    runs on a HIP stream of its own: a single fit is then a chain of dependent few-microsecond launches, and several
    chains interleave on the device.  batch_only: None (nothing run) unless the plans take the small-graph schedule -- the
    caller asked for no concurrency (n_jobs = 1) and gets the shared launches only where they are free.  The graph is uploaded once and shared; one host thread per
    stream issues the launches.  Returns
    [(G, S)] in run order; results are identical to sequential runs (each plan is deterministic)."""
    from ..._engine import upload_graph
    obj_types = list(obj_types)
    n_obj = count_objects(obj_types, R)
    rt = nat.get_runtime()
    rel_list, theta_list = upload_graph(flatten_relations(R, M), flatten_thetas(Theta), dtype, rt)
    own_streams = hasattr(rt.mem, 'new_stream')
    out = []
    for k0 in range(0, len(G0_list), max(int(n_streams), 1)):
        plans = []
        try:
            for G0 in G0_list[k0:k0 + max(int(n_streams), 1)]:
                plan = DevicePlan(obj_types, n_obj, obj_type2rank, rel_list, theta_list, variant, dtype=dtype,
                                  engine=engine, stream=rt.mem.new_stream() if own_streams else None)
                plans.append(plan)
                if batch_only and not plan.batchable():        # (decided on the first plan, before a second workspace exists)
                    return None
                plan.set_factors({t: G0[t, t] for t in obj_types}, sync=False)
            rt.mem.synchronize()                               # the uploads went out on the plans' own streams
            if len(plans) > 1 and DevicePlan.iterate_batch(plans, max_iter):
                pass      # a small graph: every launch served all the restarts of the batch (skf_iterate_batch), one stream
            elif own_streams and len(plans) > 1:
                # one host thread per stream feeds the launches (ctypes releases the GIL inside skf_iterate;
                # a plan is only ever touched by its own thread); the streams run side by side
                import threading
                errors = []

                def drive(plan):
                    try:
                        if getattr(rt.mem, 'device', None) is not None and hasattr(rt.mem, 'torch'):
                            rt.mem.torch.cuda.set_device(rt.mem.device)      # the current device is per thread
                        plan.iterate(max_iter)
                    except Exception as exc:          # surfaced after the join
                        errors.append(exc)
                threads = [threading.Thread(target=drive, args=(plan,)) for plan in plans]
                for th in threads:
                    th.start()
                for th in threads:
                    th.join()
                if errors:
                    raise errors[0]
            else:
                for plan in plans:
                    plan.iterate(max_iter)
            rt.mem.synchronize()
            fetched = [plan.fetch_results(obj_types, len(rel_list)) for plan in plans]
            rt.mem.synchronize()                               # one synchronisation for the read-back of every plan
            for plan, f in zip(plans, fetched):
                out.append(_unpack(plan.read_results(f), obj_types, rel_list))
        finally:
            for plan in plans:
                plan.close()
    return out


def run_fit(variant, R, M, Theta, obj_types, obj_type2rank, max_iter, init_type, stopping,
            stopping_system, verbose, compute_err, callback, random_state, dtype, G0, engine):
    """Shared driver of dfmf / dfmc: the body of the reference loops (_dfmf.py:212-322,
    _dfmc.py:270-392) with the arithmetic on the device."""
    logging.basicConfig(format="%(asctime)s %(levelname)s: %(message)s", level=50 - verbose)
    obj_types = list(obj_types)
    n_obj = count_objects(obj_types, R)
    if G0 is None:
        R_first = {k: np.asarray(v[0], dtype=float) for k, v in R.items()}
        G0 = initialize(obj_types, n_obj, obj_type2rank, R_first, init_type, _as_rs(random_state))
    rel_list = flatten_relations(R, M)
    plan = DevicePlan(obj_types, n_obj, obj_type2rank, rel_list, flatten_thetas(Theta), variant,
                      dtype=dtype, engine=engine)
    try:
        plan.set_factors({t: G0[t, t] for t in obj_types})
        if not (callback or stopping or stopping_system or compute_err):
            plan.iterate(max_iter)              # whole loop device-resident, no host sync
        else:
            _host_loop(lambda: plan.iterate(1), lambda idx: [plan.relation_sqerr(k) for k in idx], rel_list,
                       max_iter, stopping, stopping_system, compute_err,
                       (lambda it: callback(*(_collect(plan, obj_types, rel_list) + (it,)))) if callback else None)
        return _collect(plan, obj_types, rel_list)
    finally:
        plan.close()


def dfmf(R, Theta, obj_types, obj_type2rank, max_iter=10, init_type="random_vcol",
         stopping=None, stopping_system=None, verbose=0, compute_err=False, callback=None,
         random_state=None, n_jobs=1, dtype='f64', G0=None, engine=None, shard=None):
    """Data fusion by matrix factorization -- drop-in for reference ``dfmf`` (_dfmf.py:127).
    ``shard='relations'`` (with an initialised torch.distributed group) partitions the relations
    of this ONE fit over the ranks."""
    if shard in ('relations', 'rows', 'owned'):
        logging.basicConfig(format="%(asctime)s %(levelname)s: %(message)s", level=50 - verbose)
        fit = {'relations': run_fit_sharded, 'rows': run_fit_rows, 'owned': run_fit_owned}[shard]
        return fit(nat.SKF_DFMF, R, None, Theta, obj_types, obj_type2rank, max_iter,
                   init_type, random_state, dtype, G0, engine, stopping, stopping_system, compute_err, callback)
    return run_fit(nat.SKF_DFMF, R, None, Theta, obj_types, obj_type2rank, max_iter, init_type,
                   stopping, stopping_system, verbose, compute_err, callback, random_state,
                   dtype, G0, engine)


def _transform_graph(R_ij, target_obj_type, G):
    """(types, n_obj, n_t) of a fold-in: the target + every partner type of the new relations (reference _dfmf.py:342-350)."""
    t = target_obj_type
    sizes = [R_ij[i, j][0].shape[0 if t == i else 1] for i, j in R_ij]
    if len(set(sizes)) > 1:
        from ..base import DataFusionError
        raise DataFusionError("Target object type: %s size mismatch" % t)
    types = [t]
    for (i, j) in R_ij:
        for o in (i, j):
            if o not in types:
                types.append(o)
    n_obj = {t: sizes[0]}
    for o in types[1:]:
        n_obj[o] = G[o, o].shape[0]
    return types, n_obj, sizes[0]


def transform_runs(R_ij, Theta_i, target_obj_type, obj_type2rank, models, max_iter=10, init_type="random_c",
                   random_state=None, dtype='f64', G0=None, engine=None):
    """The fold-ins of ONE set of new relations into the models of several restarts -- `models` = [(G, S)] per restart, as
    `transform` takes them -- in shared launches: the reference runs `transform` once per restart on joblib workers
    (dfmf.py:191-199).  The new relations and constraints go to the device ONCE, every restart gets a plan with its frozen
    factors / backbones, and one launch per iteration serves all of them (skf_iterate_batch; plans that do not batch --
    a constraint on the target type -- are iterated one after the other, still on the shared uploads).  G0 of restart k is
    drawn k-th from `random_state` (the order in which the per-restart calls would draw).  Returns [G_target] per restart."""
    from ..._engine import upload_graph
    t = target_obj_type
    types, n_obj, n_t = _transform_graph(R_ij, t, models[0][0])
    rs = _as_rs(random_state)
    if G0 is None:
        R_first = {k: np.asarray(v[0], dtype=float) for k, v in R_ij.items()}
        G0 = [initialize([t], {t: n_t}, obj_type2rank, R_first, init_type, rs)[t, t] for _ in models]
    rt = nat.get_runtime()
    rel_list, theta_list = upload_graph(flatten_relations(R_ij), flatten_thetas(Theta_i), dtype, rt)
    plans = []
    try:
        for (G, S), g0 in zip(models, G0):
            plan = DevicePlan(types, n_obj, obj_type2rank, rel_list, theta_list, nat.SKF_TRANSFORM, dtype=dtype, target=t,
                              engine=engine)
            plans.append(plan)
            plan.set_factors(dict([(t, g0)] + [(o, G[o, o]) for o in types[1:]]), sync=False)
            seen = {}
            for k, (i, j, _, _) in enumerate(rel_list):
                l = seen.get((i, j), 0)
                seen[i, j] = l + 1
                plan.set_backbone(k, S[i, j][l])
        rt.mem.synchronize()
        if not (len(plans) > 1 and plans[0].batchable() and DevicePlan.iterate_batch(plans, max_iter)):
            for plan in plans:
                plan.iterate(max_iter)
        rt.mem.synchronize()
        return [plan.get_factor(t) for plan in plans]
    finally:
        for plan in plans:
            plan.close()


def transform(R_ij, Theta_i, target_obj_type, obj_type2rank, G, S, max_iter=10,
              init_type="random_c", stopping=None, stopping_system=None, verbose=0,
              compute_err=False, callback=None, random_state=None, dtype='f64', G0=None,
              engine=None):
    """Fold new objects of ``target_obj_type`` into a fitted latent space -- drop-in for
    reference ``transform`` (_dfmf.py:330-458): only the target factor moves, ``G`` of the other
    types and ``S`` (1-element lists per pair) stay frozen.  callback(G_i, iter)."""
    logging.basicConfig(format="%(asctime)s %(levelname)s: %(message)s", level=50 - verbose)
    t = target_obj_type
    sizes = [R_ij[i, j][0].shape[0 if t == i else 1] for i, j in R_ij]
    if len(set(sizes)) > 1:
        from ..base import DataFusionError
        raise DataFusionError("Target object type: %s size mismatch" % t)
    n_t = sizes[0]
    if G0 is None:
        R_first = {k: np.asarray(v[0], dtype=float) for k, v in R_ij.items()}
        G0 = initialize([t], {t: n_t}, obj_type2rank, R_first, init_type,
                        _as_rs(random_state))[t, t]
    # object types taking part: the target + every partner type of the new relations
    types = [t]
    for (i, j) in R_ij:
        for o in (i, j):
            if o not in types:
                types.append(o)
    n_obj = {t: n_t}
    for o in types[1:]:
        n_obj[o] = G[o, o].shape[0]
    rel_list = flatten_relations(R_ij)
    plan = DevicePlan(types, n_obj, obj_type2rank, rel_list, flatten_thetas(Theta_i),
                      nat.SKF_TRANSFORM, dtype=dtype, target=t, engine=engine)
    try:
        plan.set_factor(t, G0)
        for o in types[1:]:
            plan.set_factor(o, G[o, o])
        seen = {}
        for k, (i, j, _, _) in enumerate(rel_list):
            l = seen.get((i, j), 0)
            seen[i, j] = l + 1
            plan.set_backbone(k, S[i, j][l])
        if not (callback or stopping or stopping_system or compute_err):
            plan.iterate(max_iter)
        else:
            # reference _dfmf.py:367-376, 433-450: the system error of the NEW relations per iteration and
            # `stopping_system` on its change.  (`stopping` reads an undefined name there -- a NameError at the
            # third iteration; here it is the error change of the named relation, as in dfmf().)
            _host_loop(lambda: plan.iterate(1), lambda idx: [plan.relation_sqerr(k) for k in idx], rel_list,
                       max_iter, stopping, stopping_system, compute_err,
                       (lambda it: callback(plan.get_factor(t), it)) if callback else None)
        return plan.get_factor(t)
    finally:
        plan.close()
