"""Ownership-aligned row sharding (SKF_OPT_OWNED_ROWS, `shard='owned'`; SURVEY.md 8e second row; replaces the reference's
per-block joblib tasks, _dfmf.py:69-73, _dfmc.py:341-345) on the CPU: the real plan / schedule / kernel sources in the host
emulator, the ranks of a group as threads of this process (helpers.ThreadGroup) driving skf_iterate_dist.  The same cases run
over real gloo process groups of 2 and 3 ranks in test_distributed_gloo.py and on the hardware in test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import skfusion_amd._native as nat
from skfusion_amd.fusion.decomposition import _dfmf, _dfmc
from emul.runtime import emulated_runtime, use_runtime
from oracle import dfmf_oracle as orc
from helpers import (golden, probe_graph, movielens_style_graph, g0_from, relerr, fit_owned, C5_TYPES, C5_RELATIONS)
from skfusion_amd._engine import owned_rows            # noqa: F401  (also used through this module by the GPU suite)


@pytest.fixture(scope='module', autouse=True)
def emul():
    from skfusion_amd._engine import split_clamps
    with use_runtime(emulated_runtime()) as rt:
        yield rt
        assert split_clamps(rt) == 0        # no split-K launch of the module outgrew the scratch its plan sized


PAIRS = [('t1', 't2', 0), ('t1', 't2', 1), ('t1', 't3', 0), ('t2', 't3', 0)]


@pytest.mark.parametrize('variant', ['dfmf', 'dfmc'])
def test_owned_rows_reproduce_the_reference_golden_on_2_and_3_ranks(variant):
    """Probe graph (multi-relation, negative values, dense constraints, masks, a None mask): every rank of 2 and of 3
    reproduces iteration 10 of the reference golden to 1e-9, the bar of the single-device engine; what the ranks sent is
    what skf_exchange_bytes says; two calls of five iterations are one call of ten."""
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank = probe_graph(z)
    for size in (2, 3):
        out, grp, said = fit_owned(variant, R, M, Theta, types, rank, g0_from(z, variant + '/', types), 10, size, calls=2)
        for G, S in out:
            for t in types:
                assert relerr(G[t, t], z['%s/G_%s_it9' % (variant, t)]) < 1e-9
            for i, j, l in PAIRS:
                assert relerr(S[i, j][l], z['%s/S_%s_%s_%d_it9' % (variant, i, j, l)]) < 1e-9
        assert len(set(said)) == 1 and abs(grp.bytes_sent_per_rank() / 10.0 - said[0]) <= 1.0


def test_owned_rows_c5_movielens_style_dfmc():
    """BASELINE config 5 (scaled): six types, six relations, 98 % of the ratings unknown, constraints on two types, over 2
    and 3 ranks: iteration 2 of the reference golden."""
    z = golden('c5_movielens_scaled.npz')
    R, M, Theta, types, rank = movielens_style_graph()
    for size in (2, 3):
        out, _, _ = fit_owned('dfmc', R, M, Theta, types, rank, g0_from(z, 'dfmc/', types), 2, size)
        for G, S in out:
            for t in C5_TYPES:
                assert relerr(G[t, t], z['dfmc/G_%s_it1' % t]) < 1e-9
            for i, j, _, _ in C5_RELATIONS:
                assert relerr(S[i, j][0], z['dfmc/S_%s_%s_0_it1' % (i, j)]) < 1e-9


def _wide_graph(seed=11, rank=None):
    """Ranks above 64 (the three-stream schedule with the exchanges on their own stream), row counts that split at multiples
    of 64 (bf16), a masked relation, a None mask, a sparse constraint, a type that sits on the column side only."""
    rs = np.random.RandomState(seed)
    types = ['u', 'm', 'g']
    n = {'u': 200, 'm': 150, 'g': 140}
    rank = rank or {'u': 66, 'm': 70, 'g': 68}
    R = {('u', 'm'): [rs.rand(200, 150)], ('m', 'g'): [(rs.rand(150, 140) < 0.2).astype(np.float64)],
         ('u', 'g'): [rs.rand(200, 140) - 0.2]}
    M = {('u', 'm'): [rs.rand(200, 150) < 0.6], ('m', 'g'): [None], ('u', 'g'): [None]}
    Tu = 0.05 * np.eye(200)                      # (the constraint sits on the ROW type of the masked relation: its column type
    Tu[3, 7] = Tu[7, 3] = -0.01                  #  'm' then travels as bf16 rows, and the completion operand is built from them)
    Tu[190, 2] = Tu[2, 190] = 0.02
    Theta = {('u', 'u'): [Tu]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    return R, M, Theta, types, rank, G0


@pytest.mark.parametrize('variant', ['dfmf', 'dfmc'])
def test_owned_rows_wide_ranks_all_engines(variant):
    """f64 against the oracle (1e-9); f32 and bf16 against the single-device fit of the same engine (other summation order of
    the partial Gram / W / Q sums: f32 1e-5; bf16 1e-3 -- a factor entry that lands on the other side of a bf16 rounding
    boundary moves the operand by 2^-9; bf16 DFMC 5e-3: the single-device pipeline forms W of the masked relation through
    the narrower factor, (R^T G_i)^T G_j, the sharded one as G_i^T (R G_j) -- other bf16 products).  bf16: the constrained
    type ('u') gathers its f32 rows, and so do both types of a masked relation (DFMC: 'u' and 'm'); the others travel as bf16
    rows only."""
    R, M, Theta, types, rank, G0 = _wide_graph()
    its = 3
    if variant == 'dfmf':
        Go, So = orc.dfmf(R, Theta, types, rank, max_iter=its, G0=G0)
    else:
        Go, So = orc.dfmc(R, M, Theta, types, rank, max_iter=its, G0=G0)
    for dtype, tol in (('f64', 1e-9), ('f32', 1e-5), ('bf16', 1e-3 if variant == 'dfmf' else 5e-3)):
        if dtype == 'f64':
            Gs, Ss = Go, So
        elif variant == 'dfmf':
            Gs, Ss = _dfmf.dfmf(R, Theta, types, rank, max_iter=its, G0=G0, dtype=dtype)
        else:
            Gs, Ss = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=its, G0=G0, dtype=dtype)
        for size in ((2,) if dtype == 'f64' else (3,)):
            out, grp, said = fit_owned(variant, R, M, Theta, types, rank, G0, its, size, dtype=dtype)
            for G, S in out:
                for t in types:
                    assert relerr(G[t, t], Gs[t, t]) < tol, (dtype, size, t)
                for k in Ss:
                    assert relerr(S[k][0], Ss[k][0]) < 10 * tol, (dtype, size, k)
            for t in types:                       # every rank ends with the same factors, bit for bit
                for G, _ in out[1:]:
                    np.testing.assert_array_equal(G[t, t], out[0][0][t, t])
            # per iteration what skf_exchange_bytes says; bf16: plus ONE gather of the f32 rows of the types that travel as
            # bf16 rows at the end of the call ('u' carries a constraint and gathers its f32 rows every iteration; in the DFMC
            # fit so does 'm', the other type of the masked relation)
            final = 0.0
            if dtype == 'bf16':
                final = (size - 1) / float(size) * sum(owned_rows(dtype, G0[t, t].shape[0], 0, size)[2] * size * rank[t] * 4
                                                       for t in (('g',) if variant == 'dfmc' else ('m', 'g')))
            assert abs(grp.bytes_sent_per_rank() - its * said[0] - final) <= float(its)


@pytest.mark.parametrize('variant', ['dfmf', 'dfmc'])
def test_owned_rows_mixed_ranks(variant):
    """Ranks on both sides of 64 in one graph (config 5 has 16 ... 256): the three-stream schedule is kept, the small
    types' c x c work runs beside the contractions.  f64 against the oracle, f32 / bf16 against the single-device fit."""
    R, M, Theta, types, rank, G0 = _wide_graph(rank={'u': 72, 'm': 16, 'g': 64})
    its = 3
    if variant == 'dfmf':
        Go, So = orc.dfmf(R, Theta, types, rank, max_iter=its, G0=G0)
    else:
        Go, So = orc.dfmc(R, M, Theta, types, rank, max_iter=its, G0=G0)
    for dtype, tol in (('f64', 1e-9), ('f32', 1e-5), ('bf16', 5e-3)):
        if dtype == 'f64':
            Gs, Ss = Go, So
        elif variant == 'dfmf':
            Gs, Ss = _dfmf.dfmf(R, Theta, types, rank, max_iter=its, G0=G0, dtype=dtype)
        else:
            Gs, Ss = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=its, G0=G0, dtype=dtype)
        out, _, _ = fit_owned(variant, R, M, Theta, types, rank, G0, its, 2, dtype=dtype)
        for G, S in out:
            for t in types:
                assert relerr(G[t, t], Gs[t, t]) < tol, (dtype, t)
            for k in Ss:
                assert relerr(S[k][0], Ss[k][0]) < 10 * tol, (dtype, k)


def _bare_plan(lib, sizes, ranks, rels, dtype, part, flags):
    """A plan that is created but never bound (no device memory at all): enough for skf_exchange_bytes at any size."""
    tdesc = (nat.TypeDesc * len(sizes))()
    for k, (n, c) in enumerate(zip(sizes, ranks)):
        tdesc[k].n_obj, tdesc[k].rank = n, c
    rdesc = (nat.RelationDesc * len(rels))()
    for k, (i, j) in enumerate(rels):
        b, cnt, ch = C.c_int64(), C.c_int64(), C.c_int64()
        assert lib.skf_owned_rows(dtype, sizes[i], part[0], part[1], C.byref(b), C.byref(cnt), C.byref(ch)) == 0
        rdesc[k].row_type, rdesc[k].col_type = i, j
        rdesc[k].data, rdesc[k].ld = 256, sizes[j]            # (never dereferenced: the plan is not bound)
        rdesc[k].row_begin, rdesc[k].n_rows = b.value, cnt.value
        if cnt.value == 0:
            rdesc[k].flags = nat.SKF_REL_ABSENT
    opt = nat.Options(dtype, nat.SKF_DFMF, -1, nat.SKF_ENGINE_MFMA, part[0], part[1], flags)
    h = nat._P()
    hdesc = (nat.ThetaDesc * 1)()
    rc = lib.skf_plan_create(len(sizes), tdesc, len(rels), rdesc, 0, hdesc, C.byref(opt), C.byref(h))
    assert rc == 0, lib.skf_last_error()
    return h


def test_exchange_bytes_of_config_3_on_8_ranks(emul):
    """BASELINE config 3 (50k x 100k / 50k x 40k / 100k x 40k, ranks 128 / 256 / 256) over 8 ranks, bf16 engine: a rank sends
    at most 200 MB per iteration (<= 200 MB; the E / D exchange of the cost-balanced row blocks sent 605 MB)
    -- 98.6 MB of partial Q (f32, reduce-scatter), 74 MB of bf16 factor rows (all-gather), 3.8 MB of c x c sums -- and every
    rank holds 1/8 of every relation within 2.5 %."""
    lib = emul.lib
    sizes, ranks, rels = [50000, 100000, 40000], [128, 256, 256], [(0, 1), (0, 2), (1, 2)]
    seen = []
    for r in range(8):
        h = _bare_plan(lib, sizes, ranks, rels, nat.SKF_BF16, (r, 8), nat.SKF_OPT_OWNED_ROWS)
        b = C.c_size_t()
        assert lib.skf_exchange_bytes(h, 8, C.byref(b)) == 0
        seen.append(b.value)
        lib.skf_plan_destroy(h)
    assert len(set(seen)) == 1
    assert 150e6 < seen[0] <= 200e6, seen[0]
    rows = []
    for n in sizes:
        for r in range(8):
            b, cnt, ch = C.c_int64(), C.c_int64(), C.c_int64()
            assert lib.skf_owned_rows(nat.SKF_BF16, n, r, 8, C.byref(b), C.byref(cnt), C.byref(ch)) == 0
            assert b.value % 256 == 0 and ch.value % 256 == 0 and cnt.value <= ch.value
            rows.append((n, r, b.value, cnt.value))
            assert cnt.value <= 1.025 * n / 8.0
    for n in sizes:                               # the ranges tile the type
        mine = [x for x in rows if x[0] == n]
        assert mine[0][2] == 0 and sum(x[3] for x in mine) == n
        for a, b in zip(mine, mine[1:]):
            assert a[2] + a[3] == b[2] or b[3] == 0
    # the same graph in the f32 engine: factor rows travel as f32 (2 x the bf16 gather)
    h = _bare_plan(lib, sizes, ranks, rels, nat.SKF_F32, (0, 8), nat.SKF_OPT_OWNED_ROWS)
    b = C.c_size_t()
    assert lib.skf_exchange_bytes(h, 8, C.byref(b)) == 0
    lib.skf_plan_destroy(h)
    assert 240e6 < b.value < 260e6


def test_owned_rows_abi_errors(emul):
    """A block that is not the owned range, a stage call on such a plan, a communicator of another shape."""
    from skfusion_amd._engine import DevicePlan, owned_rows
    rs = np.random.RandomState(2)
    types, n, rank = ['a', 'b'], {'a': 40, 'b': 30}, {'a': 4, 'b': 3}
    Rab = rs.rand(40, 30)
    b0, c0, _ = owned_rows('f64', 40, 0, 2)
    assert (b0, c0) == (0, 20) and owned_rows('f64', 40, 1, 2)[:2] == (20, 20)
    assert owned_rows('bf16', 40, 1, 2)[:2] == (40, 0) and owned_rows('bf16', 50000, 7, 8) == (44800, 5200, 6400)
    good = dict(row_begin=0, n_rows=20, absent=False, masked=False)
    bad = dict(row_begin=0, n_rows=24, absent=False, masked=False)
    with pytest.raises(nat.SkfNativeError):
        DevicePlan(types, n, rank, [('a', 'b', Rab[:24], None, bad)], [], nat.SKF_DFMF, part=(0, 2), owned=True)
    plan = DevicePlan(types, n, rank, [('a', 'b', Rab[:20], None, good)], [], nat.SKF_DFMF, part=(0, 2), owned=True)
    for t in types:
        plan.set_factor(t, rs.rand(n[t], rank[t]))
    with pytest.raises(nat.SkfNativeError):
        plan.stage(nat.SKF_STAGE_CONTRACT)
    with pytest.raises(nat.SkfNativeError):
        plan.iterate(1)
    with pytest.raises(nat.SkfNativeError):           # no communicator yet
        plan.iterate_dist(1)
    plan.attach_null_comm(0, 3)
    with pytest.raises(nat.SkfNativeError):           # rank 0 of 3 is not part 0 of 2
        plan.iterate_dist(1)
    plan.close()
    plan = DevicePlan(types, n, rank, [('a', 'b', Rab[:20], None, good)], [], nat.SKF_DFMF, part=(0, 2), owned=True)
    for t in types:
        plan.set_factor(t, rs.rand(n[t], rank[t]))
    plan.attach_null_comm(0, 2)
    plan.iterate_dist(2)                              # timing vehicle: runs, exchanges skipped
    assert np.isfinite(plan.get_factor('a')).all()
    plan.close()


def test_owned_rows_dense_constraint_in_the_bf16_engine():
    """A DENSE constraint (more than n^2 / 16 non-zeros) under row ownership in the bf16 engine: a rank multiplies ITS rows of
    the bf16 halves of Theta with the stored G^T (`theta_terms_rows`); 2 ranks (rows split at 128) against the single-device
    fit of the same engine."""
    rs = np.random.RandomState(17)
    types, n, rank = ['a', 'b'], {'a': 256, 'b': 140}, {'a': 12, 'b': 9}
    R = {('a', 'b'): [rs.rand(256, 140)]}
    Ta = 0.02 * (rs.rand(256, 256) - 0.3)
    Ta = 0.5 * (Ta + Ta.T)
    Theta = {('a', 'a'): [Ta]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    Gs, Ss = _dfmf.dfmf(R, Theta, types, rank, max_iter=3, G0=G0, dtype='bf16')
    out, _, _ = fit_owned('dfmf', R, None, Theta, types, rank, G0, 3, 2, dtype='bf16')
    for G, S in out:
        for t in types:
            assert relerr(G[t, t], Gs[t, t]) < 1e-3
        assert relerr(S['a', 'b'][0], Ss['a', 'b'][0]) < 1e-2


def test_owned_rows_rank_deficient_gram():
    """Rank 50 > 30 objects (reference tests/test_n_run.py:14): the partial Gram matrices of the owners sum to a singular
    matrix, every rank takes the rank-revealing pseudo-inverse path on the summed matrix and lands on the reference's
    iterates (the golden of the single-device engine, 1e-7) and on the oracle's reconstruction errors."""
    from helpers import rank_deficient_graph
    z = golden('rank_deficient.npz')
    R, types, rank = rank_deficient_graph(z)
    G0 = g0_from(z, 'dfmf/', types)
    Go, So = orc.dfmf(R, {}, types, rank, max_iter=2, G0=G0)
    eo = orc.relation_errors(R, Go, So)
    out, _, _ = fit_owned('dfmf', R, None, {}, types, rank, G0, 2, 2)
    for G, S in out:
        for t in types:
            assert relerr(G[t, t], z['dfmf/G_%s_it1' % t]) < 1e-7
            assert np.isfinite(G[t, t]).all()
        e = orc.relation_errors(R, G, S)
        for k in e:
            assert abs(e[k][0] - eo[k][0]) <= 1e-6 * max(1.0, eo[k][0])


@pytest.mark.parametrize('dtype,ranks,tol', [('f64', {'a': 16, 'b': 12, 'c': 8}, 1e-9), ('bf16', {'a': 128, 'b': 64, 'c': 16}, 2e-2)])
def test_owned_rows_keep_the_lists_of_known_entries(dtype, ranks, tol, monkeypatch):
    """DFMC on the known entries only (skf_known.h) UNDER row ownership: a rank keeps the lists of its own rows; the
    stored-residual pass gives its share of W (all-reduce), the row pass its rows of P S^T, the column pass its partial
    E^T G_i (reduce-scatter) and every rank -- also one without rows of the relation -- adds the dense part G_j (S^T Gram_i)
    on ITS rows of the column type.  2 and 3 ranks against the oracle (f64 1e-9; bf16: engine tolerance, v6 list kernel at
    rank 128), the summed squared errors of the ranks against the oracle's."""
    import known_cases as KC
    monkeypatch.setenv('SKF_DFMC_SPARSE', '1')
    n = {'a': 200, 'b': 140, 'c': 130}
    types, rels, thetas, G0 = KC.masked_graph(n, ranks, 0.05, seed=3)
    R = {(i, j): [m] for i, j, m, _ in rels}
    M = {(i, j): [mask] for i, j, _, mask in rels}
    Theta = {(t, t): [th] for t, th in thetas}
    G0d = {(t, t): G0[t] for t in types}
    its = 4
    Go, So = orc.dfmc(R, M, Theta, types, ranks, max_iter=its, G0=G0d)
    # the oracle's errors on ITS working copy (masked entries = the completion of the last iteration)
    Rw = {k: [m.copy() for m in v] for k, v in R.items()}
    for k, masks in M.items():
        if masks[0] is not None:
            Rw[k][0][masks[0]] = (Go[k[0], k[0]] @ So[k][0] @ Go[k[1], k[1]].T)[masks[0]]
    for size in (2, 3):
        sq = []
        out, grp, said = fit_owned('dfmc', R, M, Theta, types, ranks, G0d, its, size, dtype=dtype, sqerr=sq)
        for G, S in out:
            for t in types:
                assert relerr(G[t, t], Go[t, t]) < tol, (dtype, size, t)
        if dtype == 'f64':
            tot = np.sum(np.array(sq), axis=0)
            # (the engine's residual of a list relation refers to the completion of the iteration BEFORE the last update of S:
            #  compare the unmasked relation exactly and the masked ones through their known entries' share)
            eo = orc.relation_errors(R, Go, So)
            k_unmasked = [k for k, (i, j, m, mask) in enumerate(rels) if mask is None][0]
            i, j = rels[k_unmasked][0], rels[k_unmasked][1]
            assert abs(np.sqrt(tot[k_unmasked]) - eo[i, j][0]) < 1e-8 * eo[i, j][0]
            assert np.isfinite(tot).all() and (tot > 0).all()


def test_comm_info_and_launch_count(emul):
    """Round 5: what a communicator says about itself (skf_comm_info: the `strong` record of bench.py quotes it) and the launch
    counter behind `launches_per_step` (skf_launch_count) -- on the emulated runtime."""
    from skfusion_amd._engine import DevicePlan, owned_rows, launch_count
    rs = np.random.RandomState(1)
    types, n, rank = ['a', 'b'], {'a': 40, 'b': 30}, {'a': 4, 'b': 3}
    Rab = rs.rand(40, 30)
    blk = dict(absent=False, row_begin=0, n_rows=20, masked=False)
    plan = DevicePlan(types, n, rank, [('a', 'b', Rab[:20], None, blk)], [], nat.SKF_DFMF, part=(0, 2), owned=True)
    try:
        assert plan.comm_info() is None
        plan.attach_null_comm(0, 2)
        assert plan.comm_info() == {'rank': 0, 'world': 2, 'transport': 'null', 'transport_ranks': 0}
        plan.attach_callback_comm(0, 2, lambda op, view, count, r, w: None)
        assert plan.comm_info() == {'rank': 0, 'world': 2, 'transport': 'callback', 'transport_ranks': 2}
    finally:
        plan.close()
    plan = DevicePlan(types, n, rank, [('a', 'b', Rab, None)], [], nat.SKF_DFMF)
    try:
        plan.attach_single_comm()
        assert plan.comm_info() == {'rank': 0, 'world': 1, 'transport': 'single', 'transport_ranks': 1}
        for t in types:
            plan.set_factor(t, rs.rand(n[t], rank[t]) + 0.1)
        before = launch_count()
        plan.iterate(2)
        per_iteration = (launch_count() - before) / 2.0
        assert per_iteration == 3.0                      # the three-launch schedule of small graphs
    finally:
        plan.close()
