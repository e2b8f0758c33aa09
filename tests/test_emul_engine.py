"""The whole device engine (plan + launch schedule + kernels) on the CPU via the host SIMT
emulator, compared with the golden vectors of the reference and with the oracle.  The same
comparisons run on the hardware in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import skfusion_amd._native as nat
from skfusion_amd.fusion.decomposition import _dfmf, _dfmc
from emul.runtime import emulated_runtime, use_runtime
from oracle import dfmf_oracle as orc
from helpers import (golden, readme_graph, probe_graph, rank_deficient_graph, dicty_graph, g0_from,
                     Snapshots, compare_snapshots, relerr, TYPES)


@pytest.fixture(scope='module', autouse=True)
def emul():
    from skfusion_amd._engine import split_clamps
    with use_runtime(emulated_runtime()) as rt:
        yield rt
        assert split_clamps(rt) == 0        # no split-K launch of the module outgrew the scratch its plan sized


@pytest.mark.parametrize('small_chain', ['fused', 'on', 'off'])
@pytest.mark.parametrize('engine', [nat.SKF_ENGINE_MFMA, nat.SKF_ENGINE_VALU])
def test_c1_readme_f64_matches_reference_golden(engine, small_chain, monkeypatch):
    """(small_chain off: the c x c algebra through the generic GEMM launches and the blocked Cholesky
    inverse, as for ranks above 64; on: the one-workgroup kernels for small ranks on the staged
    schedule; fused: the job-table schedule for small graphs, skf_small.h.)"""
    if small_chain == 'on':
        monkeypatch.setenv('SKF_NO_SMALL_FUSED', '1')
    if small_chain == 'off':
        monkeypatch.setenv('SKF_NO_SMALL_CHAIN', '1')
        monkeypatch.setenv('SKF_CHOL_NO_SMALL', '1')
    z = golden('c1_readme_dfmf.npz')
    R, types, rank = readme_graph()
    snaps = Snapshots((0, 1, 9))
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=10, callback=snaps,
                      G0=g0_from(z, 'random_vcol/', types), dtype='f64', engine=engine)
    worst = compare_snapshots(z, 'random_vcol/', snaps.snap, 1e-10)
    assert worst < 1e-10


def test_c1_readme_device_resident_loop_f64_and_f32():
    z = golden('c1_readme_dfmf.npz')
    R, types, rank = readme_graph()
    G0 = g0_from(z, 'random/', types)
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=10, G0=G0, dtype='f64')   # one skf_iterate(10)
    for t in types:
        assert relerr(G[t, t], z['random/G_%s_it9' % t]) < 1e-9
    for (i, j) in R:
        assert relerr(S[i, j][0], z['random/S_%s_%s_0_it9' % (i, j)]) < 1e-9
    # fp32 engine (SURVEY 8d tolerances: 1e-4 on G, 1e-5 on the reconstruction error); the
    # 30- and 100-iteration versions of this check run on the GPU (tests/test_gpu_parity.py)
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=12, G0=G0, dtype='f32')
    Go, So = orc.dfmf(R, {}, types, rank, max_iter=12, G0=G0)
    for t in types:
        assert relerr(G[t, t], Go[t, t]) < 1e-4
    for k in So:
        assert relerr(S[k][0], So[k][0]) < 1e-3
    e, eo = orc.relation_errors(R, G, S), orc.relation_errors(R, Go, So)
    for k in e:
        assert abs(e[k][0] - eo[k][0]) / eo[k][0] < 1e-5


def test_seeded_initialisers_match_reference_stream():
    """Without G0 the host initialisers must consume the RandomState like the reference."""
    z = golden('c1_readme_dfmf.npz')
    R, types, rank = readme_graph()
    for init in ('random', 'random_c', 'random_vcol'):
        snaps = Snapshots((0,))
        _dfmf.dfmf(R, {}, types, rank, max_iter=1, init_type=init, callback=snaps,
                   random_state=np.random.RandomState(0), dtype='f64')
        compare_snapshots(z, init + '/', snaps.snap, 1e-9)
    with pytest.raises(KeyError):
        _dfmf.dfmf(R, {}, types, rank, max_iter=1, init_type='nope',
                   random_state=np.random.RandomState(0))


@pytest.mark.parametrize('schedule', ['fused', 'staged'])
def test_probe_graph_dfmf_theta_multirelation_negative_values(schedule, monkeypatch):
    if schedule == 'staged':
        monkeypatch.setenv('SKF_NO_SMALL_FUSED', '1')
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank = probe_graph(z)
    snaps = Snapshots((0, 1, 9, 29))
    _dfmf.dfmf(R, Theta, types, rank, max_iter=30, callback=snaps, G0=g0_from(z, 'dfmf/', types))
    compare_snapshots(z, 'dfmf/', snaps.snap, 1e-9)


def test_probe_graph_dfmc_masks_and_inputs_untouched():
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank = probe_graph(z)
    keep = {k: [m.copy() for m in v] for k, v in R.items()}
    snaps = Snapshots((0, 1, 9, 29))
    _dfmc.dfmc(R, M, Theta, types, rank, max_iter=30, callback=snaps, G0=g0_from(z, 'dfmc/', types))
    compare_snapshots(z, 'dfmc/', snaps.snap, 1e-9)
    for k in R:
        for a, b in zip(R[k], keep[k]):
            np.testing.assert_array_equal(a, b)


def test_c5_movielens_style_dfmc_two_iterations():
    """BASELINE config 5 (scaled): 6 relations, 98 % masked ratings, lam*I and sparse negative Theta."""
    from helpers import movielens_style_graph
    z = golden('c5_movielens_scaled.npz')
    R, M, Theta, types, rank = movielens_style_graph()
    snaps = Snapshots((0, 1))
    _dfmc.dfmc(R, M, Theta, types, rank, max_iter=2, callback=snaps, G0=g0_from(z, 'dfmc/', types))
    assert compare_snapshots(z, 'dfmc/', snaps.snap, 1e-9) < 1e-9


@pytest.mark.parametrize('variant', ['dfmf', 'dfmc'])
def test_rank_deficient_gram(variant):
    z = golden('rank_deficient.npz')
    R, types, rank = rank_deficient_graph(z)
    G0 = g0_from(z, variant + '/', types)
    snaps = Snapshots((0, 1))
    if variant == 'dfmf':
        G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=4, callback=snaps, G0=G0)
        Go, So = orc.dfmf(R, {}, types, rank, max_iter=4, G0=G0)
    else:
        M = {k: [None] for k in R}
        G, S = _dfmc.dfmc(R, M, {}, types, rank, max_iter=4, callback=snaps, G0=G0)
        Go, So = orc.dfmc(R, M, {}, types, rank, max_iter=4, G0=G0)
    compare_snapshots(z, variant + '/', snaps.snap, 1e-7)
    assert all(np.isfinite(v).all() for v in G.values())
    e, eo = orc.relation_errors(R, G, S), orc.relation_errors(R, Go, So)
    for k in e:
        assert abs(e[k][0] - eo[k][0]) <= 1e-6 * max(1.0, eo[k][0])


@pytest.mark.parametrize('init', ['random_c', 'random'])
def test_transform_fold_in(init):
    z = golden('transform_readme.npz')
    G = {(t, t): z['G_%s' % t] for t in TYPES}
    S = {('t1', 't2'): [z['S_t1_t2']], ('t1', 't3'): [z['S_t1_t3']], ('t2', 't1'): [z['S_t2_t1']]}
    Rn = {('t1', 't2'): [z['new_t1_t2']], ('t1', 't3'): [z['new_t1_t3']], ('t2', 't1'): [z['new_t2_t1']]}
    rank = {'t1': 10, 't2': 20, 't3': 30}
    snaps = {}
    Gi = _dfmf.transform(Rn, {('t1', 't1'): [z['theta_t1']]}, 't1', rank, G, S, max_iter=100,
                         init_type=init, random_state=np.random.RandomState(4),
                         callback=lambda g, it: snaps.__setitem__(it, g.copy()))
    for it in (0, 9, 99):
        assert relerr(snaps[it], z['%s/G_it%d' % (init, it)]) < 1e-9
    assert relerr(Gi, z['%s/G_it99' % init]) < 1e-9
    Gi32 = _dfmf.transform(Rn, {('t1', 't1'): [z['theta_t1']]}, 't1', rank, G, S, max_iter=100,
                           G0=z[init + '/G0'], dtype='f32')
    assert relerr(Gi32, z['%s/G_it99' % init]) < 1e-4
    # bf16 engine: bf16 new relations / constraint halves / frozen G^T in the one-off contractions,
    # f32 masters (measured 2.0e-3 .. 2.1e-3 against the f64 golden)
    Gib = _dfmf.transform(Rn, {('t1', 't1'): [z['theta_t1']]}, 't1', rank, G, S, max_iter=100,
                          G0=z[init + '/G0'], dtype='bf16')
    assert relerr(Gib, z['%s/G_it99' % init]) < 1e-2


def test_c2_dicty_first_iteration_f64():
    """BASELINE config 2 inputs; the full 100-iteration run is a GPU test (emulation is slow)."""
    z = golden('c2_dicty.npz')
    R, Theta, types, rank = dicty_graph()
    G0 = g0_from(z, 'dfmf/', types)
    snaps = Snapshots((0,))
    G, S = _dfmf.dfmf(R, Theta, types, rank, max_iter=1, callback=snaps, G0=G0)
    compare_snapshots(z, 'dfmf/', snaps.snap, 1e-9)
    Go, So = orc.dfmf(R, Theta, types, rank, max_iter=1, G0=G0)
    for t in types:
        assert relerr(G[t, t], Go[t, t]) < 1e-9


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_batched_restarts_are_the_individual_runs(dtype):
    """skf_iterate_batch: three restarts of the README graph share every launch (the restart is a grid dimension); each
    plan ends with exactly the factors and backbones of its own skf_iterate run.  A plan that does not take the
    small-graph schedule is refused (SKF_E_STATE -> False)."""
    from skfusion_amd._engine import DevicePlan, flatten_relations
    R, types, rank = readme_graph()
    n = {'t1': 50, 't2': 100, 't3': 40}
    rel = flatten_relations(R)
    rs = np.random.RandomState(3)
    starts = [{t: rs.rand(n[t], rank[t]) for t in types} for _ in range(3)]

    def make(G0, dt=dtype):
        plan = DevicePlan(types, n, rank, rel, [], nat.SKF_DFMF, dtype=dt)
        for t in types:
            plan.set_factor(t, G0[t])
        return plan
    alone = []
    for G0 in starts:
        plan = make(G0)
        assert plan.batchable()
        plan.iterate(4)
        alone.append(([plan.get_factor(t) for t in types], [plan.get_backbone(k) for k in range(len(rel))]))
        plan.close()
    plans = [make(G0) for G0 in starts]
    assert DevicePlan.iterate_batch(plans, 3)
    plans[1].iterate(1)                                    # a batched plan goes on alone ...
    assert DevicePlan.iterate_batch([plans[2], plans[0]], 1)      # ... and in another batch, in another order
    for plan, (G, S) in zip(plans, alone):
        for t, g in zip(types, G):
            np.testing.assert_array_equal(plan.get_factor(t), g)
        for k, sk in enumerate(S):
            np.testing.assert_array_equal(plan.get_backbone(k), sk)
    other = make(starts[0], 'bf16')                        # the bf16 engine has no small-graph schedule
    assert not other.batchable() and not DevicePlan.iterate_batch([plans[0], other], 1)
    with pytest.raises(nat.SkfNativeError):
        DevicePlan.iterate_batch([plans[0], plans[0]], 1)
    for plan in plans + [other]:
        plan.close()


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_small_graph_schedule_on_an_awkward_graph(dtype, monkeypatch):
    """tests/small_cases.py: every job kind of skf_small.h on sizes that fit no tile; four-launch schedule vs the oracle
    and vs the general schedule."""
    import small_cases
    worst_o, worst_s = small_cases.check(dtype, monkeypatch)
    assert worst_o < (1e-10 if dtype == 'f64' else 2e-4)
    assert worst_s < (1e-11 if dtype == 'f64' else 2e-4)


@pytest.mark.parametrize('n_big', [2317, 1100])
def test_small_graph_schedule_with_many_shares(n_big, monkeypatch):
    """A type of a few thousand objects in the small-graph schedule: 37 / 18 Gram shares and 10 / 5 Q shares -- the share
    sums take their shares eight at a time (a tail of 5 / 2 on chain 0, more than one batch of Q shares per element) and keep
    the order of the additions: the oracle to 1e-10, the four-pivot and the one-pivot sweep bit for bit, the general schedule
    within rounding."""
    from skfusion_amd._engine import DevicePlan
    rs = np.random.RandomState(n_big)
    types = ['a', 'b', 'c']
    n = {'a': n_big, 'b': 37, 'c': 90}
    rank = {'a': 9, 'b': 5, 'c': 14}
    R = {('a', 'b'): [rs.rand(n_big, 37)], ('a', 'c'): [rs.rand(n_big, 90) - 0.2], ('b', 'c'): [rs.rand(37, 90)]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    rel = [(i, j, m, None) for (i, j), ms in R.items() for m in ms]

    def run():
        plan = DevicePlan(types, n, rank, rel, [], nat.SKF_DFMF, dtype='f64')
        for t in types:
            plan.set_factor(t, G0[t, t])
        plan.iterate(3)
        out = {t: plan.get_factor(t) for t in types}, [plan.get_backbone(k) for k in range(len(rel))]
        plan.close()
        return out
    Go, So = orc.dfmf(R, {}, types, rank, max_iter=3, G0=G0)
    G4, S4 = run()
    monkeypatch.setenv('SKF_SMALL_SWEEP4', '0')
    G1, S1 = run()
    monkeypatch.delenv('SKF_SMALL_SWEEP4')
    monkeypatch.setenv('SKF_NO_SMALL_FUSED', '1')
    Gg, Sg = run()
    monkeypatch.delenv('SKF_NO_SMALL_FUSED')
    for t in types:
        assert relerr(G4[t], Go[t, t]) < 1e-10
        np.testing.assert_array_equal(G4[t], G1[t])
        assert 0.0 < relerr(G4[t], Gg[t]) < 1e-11         # (not the same bits: the fused schedule ran)
    for a, b in zip(S4, [m for key in R for m in So[key]]):
        assert relerr(a, b) < 1e-10


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_device_squared_error_on_unaligned_shapes(dtype):
    """skf_relation_sqerr sizes one partial per workgroup of the tile the product runs on: the dicty relations (1219 x 116
    and 1219 x 282 at ranks 50 / 15 / 5: nothing divisible by the vector widths) take the small tile with run-time
    staging modes -- the device value must still be the host's."""
    from skfusion_amd._engine import DevicePlan, flatten_relations, flatten_thetas
    z = golden('c2_dicty.npz')
    R, Theta, types, rank = dicty_graph()
    G0 = g0_from(z, 'dfmf/', types)
    n = {'gene': R['gene', 'go'][0].shape[0], 'go': R['gene', 'go'][0].shape[1], 'exc': R['gene', 'exc'][0].shape[1]}
    rel = flatten_relations(R)
    plan = DevicePlan(types, n, rank, rel, flatten_thetas(Theta), nat.SKF_DFMF, dtype=dtype)
    for t in types:
        plan.set_factor(t, G0[t, t])
    plan.iterate(2)
    G = {t: plan.get_factor(t).astype(np.float64) for t in types}
    for k in range(len(rel)):
        i, j = rel[k][0], rel[k][1]
        host = np.linalg.norm(R[i, j][0] - G[i] @ plan.get_backbone(k).astype(np.float64) @ G[j].T)
        dev = float(np.sqrt(plan.relation_sqerr(k)))
        assert abs(dev - host) / host < (1e-10 if dtype == 'f64' else 2e-5)
    plan.close()


def test_bf16_engine_against_oracle_on_bf16_rounded_relations():
    """SKF_BF16: bf16 R / R^T / G^T feed the relation contractions, everything else as the f32
    engine.  Oracle run on the bf16-rounded relations (exactly representable in f64) so that
    only the arithmetic differs; tolerance on the reconstruction RMSE: 1e-2 relative
    (SURVEY.md 8d), on G: 2e-2."""
    R, types, rank = readme_graph()
    Rb = {k: [nat.from_bf16_bits(nat.to_bf16_bits(v[0])).astype(np.float64)] for k, v in R.items()}
    z = golden('c1_readme_dfmf.npz')
    G0 = g0_from(z, 'random/', types)
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=8, G0=G0, dtype='bf16')
    Go, So = orc.dfmf(Rb, {}, types, rank, max_iter=8, G0=G0)
    for t in types:
        assert relerr(G[t, t], Go[t, t]) < 2e-2
    e, eo = orc.relation_errors(Rb, G, S), orc.relation_errors(Rb, Go, So)
    for k in e:
        assert abs(e[k][0] - eo[k][0]) / eo[k][0] < 1e-2
    from skfusion_amd._engine import DevicePlan, flatten_relations
    rel = flatten_relations(R)
    plan = DevicePlan(types, {'t1': 50, 't2': 100, 't3': 40}, rank, rel, [], nat.SKF_DFMF, dtype='bf16')
    for t in types:
        plan.set_factor(t, G0[t, t])
    plan.iterate(2)
    Gd = {(t, t): plan.get_factor(t) for t in types}
    for k, (i, j, Rm, _) in enumerate(rel):
        # the residual pass multiplies bf16-rounded H = G_i S and G_j on the matrix cores (f32 accumulate)
        rb = lambda x: nat.from_bf16_bits(nat.to_bf16_bits(x.astype(np.float32))).astype(np.float64)
        H = Gd[i, i] @ plan.get_backbone(k)
        want_b = np.linalg.norm(Rb[i, j][0] - rb(H) @ rb(Gd[j, j]).T) ** 2
        want = np.linalg.norm(Rb[i, j][0] - H @ Gd[j, j].T) ** 2
        got = plan.relation_sqerr(k)
        assert abs(got - want_b) < 2e-5 * want_b
        assert abs(got - want) < 2e-3 * want
    plan.close()


def test_bf16_engine_odd_sizes_and_wide_rank():
    """bf16 engine on shapes that exercise every padding path: object counts that are not
    multiples of 8 / 64 (zero-padded K), a rank above 256 (several N tiles) and a rank of 1."""
    rs = np.random.RandomState(12)
    n = {'a': 77, 'b': 301, 'c': 9}
    rank = {'a': 5, 'b': 1, 'c': 3}
    R = {('a', 'b'): [rs.rand(77, 301)], ('c', 'a'): [rs.rand(9, 77)], ('b', 'c'): [rs.rand(301, 9)]}
    types = ['a', 'b', 'c']
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.1 for t in types}
    Rb = {k: [nat.from_bf16_bits(nat.to_bf16_bits(v[0])).astype(np.float64)] for k, v in R.items()}
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=6, G0=G0, dtype='bf16')
    Go, So = orc.dfmf(Rb, {}, types, rank, max_iter=6, G0=G0)
    for t in types:
        assert relerr(G[t, t], Go[t, t]) < 3e-2
    # wide rank: 300 latent dimensions on 320 objects (N = 300 > 256 -> two column tiles)
    R2 = {('a', 'b'): [rs.rand(320, 140)]}
    rank2 = {'a': 300, 'b': 20}
    G02 = {('a', 'a'): rs.rand(320, 300) + 0.1, ('b', 'b'): rs.rand(140, 20) + 0.1}
    Rb2 = {k: [nat.from_bf16_bits(nat.to_bf16_bits(v[0])).astype(np.float64)] for k, v in R2.items()}
    G, S = _dfmf.dfmf(R2, {}, ['a', 'b'], rank2, max_iter=2, G0=G02, dtype='bf16')
    Go, So = orc.dfmf(Rb2, {}, ['a', 'b'], rank2, max_iter=2, G0=G02)
    e, eo = orc.relation_errors(Rb2, G, S), orc.relation_errors(Rb2, Go, So)
    assert abs(e['a', 'b'][0] - eo['a', 'b'][0]) < 5e-2 * eo['a', 'b'][0]


def test_bf16_dfmc_masked_completion():
    """SKF_BF16 + SKF_DFMC: zeroing and completion act on both stored copies (R and R^T, bf16)."""
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank = probe_graph(z)
    Rb = {k: [nat.from_bf16_bits(nat.to_bf16_bits(m)).astype(np.float64) for m in v] for k, v in R.items()}
    G0 = g0_from(z, 'dfmc/', types)
    keep = {k: [m.copy() for m in v] for k, v in R.items()}
    G, S = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=10, G0=G0, dtype='bf16')
    Go, So = orc.dfmc(Rb, M, Theta, types, rank, max_iter=10, G0=G0)
    for t in types:
        assert relerr(G[t, t], Go[t, t]) < 5e-2
    for k in R:
        for a, b in zip(R[k], keep[k]):
            np.testing.assert_array_equal(a, b)
    # unmasked entries: reconstruction error close to the f64 oracle's
    for (i, j), mats in Rb.items():
        for l, m in enumerate(mats):
            keepm = np.ones(m.shape, bool) if M[i, j][l] is None else ~M[i, j][l]
            e = np.linalg.norm((m - G[i, i] @ S[i, j][l] @ G[j, j].T)[keepm])
            eo = np.linalg.norm((m - Go[i, i] @ So[i, j][l] @ Go[j, j].T)[keepm])
            assert abs(e - eo) < 5e-2 * eo


@pytest.mark.parametrize('variant', ['dfmf', 'dfmc'])
def test_row_block_sharding_matches_reference_golden(variant):
    """relations cut into balanced row blocks over 2 and 3 (simulated) ranks, W / Q / E / D
    summed between the stages: every rank reproduces iteration 10 of the reference golden."""
    from helpers import fit_row_blocks
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank = probe_graph(z)
    pairs = [('t1', 't2', 0), ('t1', 't2', 1), ('t1', 't3', 0), ('t2', 't3', 0)]
    for size in (2, 3):
        for G, S in fit_row_blocks(variant, R, M, Theta, types, rank, g0_from(z, variant + '/', types), 10, size):
            for t in types:
                assert relerr(G[t, t], z['%s/G_%s_it9' % (variant, t)]) < 1e-9
            for i, j, l in pairs:
                assert relerr(S[i, j][l], z['%s/S_%s_%s_%d_it9' % (variant, i, j, l)]) < 1e-9


def test_row_block_sharding_bf16_and_abi_errors():
    """bf16 row blocks (boundaries at multiples of 64; Q of a block contracts against a column
    window of the stored G^T) against the single-plan bf16 engine; misuse of the staged ABI."""
    import ctypes as C
    from helpers import fit_row_blocks
    from skfusion_amd._engine import DevicePlan, flatten_relations
    rs = np.random.RandomState(4)
    types, n, rank = ['a', 'b', 'c'], {'a': 200, 'b': 150, 'c': 70}, {'a': 6, 'b': 5, 'c': 4}
    R = {('a', 'b'): [rs.rand(200, 150)], ('a', 'c'): [rs.rand(200, 70)], ('b', 'c'): [rs.rand(150, 70)]}
    M = {('a', 'b'): [rs.rand(200, 150) > 0.7], ('a', 'c'): [None], ('b', 'c'): [None]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.1 for t in types}
    for variant in ('dfmf', 'dfmc'):
        mod = _dfmf if variant == 'dfmf' else _dfmc
        if variant == 'dfmf':
            Gs, Ss = _dfmf.dfmf(R, {}, types, rank, max_iter=3, G0=G0, dtype='bf16')
        else:
            Gs, Ss = _dfmc.dfmc(R, M, {}, types, rank, max_iter=3, G0=G0, dtype='bf16')
        for G, S in fit_row_blocks(variant, R, M, {}, types, rank, G0, 3, 2, dtype='bf16'):
            for t in types:
                assert relerr(G[t, t], Gs[t, t]) < 1e-4          # f32 partial sums in another order
            for k in Ss:
                assert relerr(S[k][0], Ss[k][0]) < 1e-4
    # ABI misuse
    rel = flatten_relations(R)
    blk = dict(row_begin=64, n_rows=64, absent=False, col_side=False, masked=False)
    plan = DevicePlan(types, n, rank, [(rel[0][0], rel[0][1], rel[0][2][64:128], None, blk)] + rel[1:], [],
                      nat.SKF_DFMF, dtype='f64')
    for t in types:
        plan.set_factor(t, G0[t, t])
    with pytest.raises(nat.SkfNativeError):           # a plan with row blocks has no skf_iterate
        plan.iterate(1)
    plan.stage(nat.SKF_STAGE_CONTRACT)
    with pytest.raises(nat.SkfNativeError):
        plan.stage(7)
    plan.close()
    bad = dict(row_begin=150, n_rows=100, absent=False, col_side=True, masked=False)
    with pytest.raises(nat.SkfNativeError):           # block outside the row type
        DevicePlan(types, n, rank, [(rel[0][0], rel[0][1], rel[0][2][:100], None, bad)] + rel[1:], [],
                   nat.SKF_DFMF, dtype='f64')
    bad = dict(row_begin=10, n_rows=64, absent=False, col_side=True, masked=False)
    with pytest.raises(nat.SkfNativeError):           # bf16 blocks start at multiples of 64
        DevicePlan(types, n, rank, [(rel[0][0], rel[0][1], rel[0][2][10:74], None, bad)] + rel[1:], [],
                   nat.SKF_DFMF, dtype='bf16')


@pytest.mark.parametrize('known', [0.6, 0.04])
def test_bf16_completion_kernel_against_f32_engine(known):
    """(known = 0.04: the known entries travel as compact per-tile lists and the completion writes whole tiles;
    0.6: more than 1/8 known, the completion blends through the mask.)
    gemm_bf16_v2_kernel<.., EPI_T_COMPLETE> (bf16 H, G_j on the matrix cores; the one stored copy of
    R written in 16-byte chunks with the known entries blended back; the mask as packed bits) against the
    f32 engine's per-element masked store on the bf16-rounded relation, on shapes that are not multiples of
    the 128 x 128 tile / of 8.  The mask arrives once as host booleans (packed on the host,
    SKF_REL_MASK_BITS) and once as device bytes (packed by the library at bind time).
    (SKF_DFMC_SPARSE=0: this is the test of the DENSE path with its completed copy; by default a relation with so few
    known entries is kept as lists of them -- test_dfmc_on_the_known_entries_only_*.)"""
    import os
    from skfusion_amd._engine import DevicePlan, DeviceMatrix, flatten_relations
    saved = os.environ.get('SKF_DFMC_SPARSE')
    os.environ['SKF_DFMC_SPARSE'] = '0'
    try:
        _bf16_completion_case(known, DevicePlan, DeviceMatrix)
    finally:
        if saved is None:
            os.environ.pop('SKF_DFMC_SPARSE')
        else:
            os.environ['SKF_DFMC_SPARSE'] = saved


def _bf16_completion_case(known, DevicePlan, DeviceMatrix):
    rs = np.random.RandomState(9)
    types, n, rank = ['a', 'b'], {'a': 203, 'b': 157}, {'a': 7, 'b': 5}
    R = {('a', 'b'): [rs.rand(203, 157)]}
    mask = rs.rand(203, 157) > known
    mask[5, :] = True            # a fully unknown row, a fully known row, a fully unknown column
    mask[6, :] = False
    mask[:, 11] = True
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.1 for t in types}
    Rb = {k: [nat.from_bf16_bits(nat.to_bf16_bits(v[0])).astype(np.float64)] for k, v in R.items()}
    Gf, Sf = _dfmc.dfmc(Rb, {('a', 'b'): [mask]}, {}, types, rank, max_iter=3, G0=G0, dtype='f32')
    G, S = _dfmc.dfmc(R, {('a', 'b'): [mask]}, {}, types, rank, max_iter=3, G0=G0, dtype='bf16')
    for t in types:                                   # measured 2.5e-4 / 4.3e-4 (G), 3.4e-3 (S) in round 1
        assert relerr(G[t, t], Gf[t, t]) < 2e-3
    assert relerr(S['a', 'b'][0], Sf['a', 'b'][0]) < 2e-2
    # the known entries are untouched: reconstruction error on them equals the oracle's
    Go, So = orc.dfmc(Rb, {('a', 'b'): [mask]}, {}, types, rank, max_iter=3, G0=G0)
    e = np.linalg.norm((Rb['a', 'b'][0] - G['a', 'a'] @ S['a', 'b'][0] @ G['b', 'b'].T)[~mask])
    eo = np.linalg.norm((Rb['a', 'b'][0] - Go['a', 'a'] @ So['a', 'b'][0] @ Go['b', 'b'].T)[~mask])
    assert abs(e - eo) < 2e-2 * eo
    # byte mask in device memory with a pitch that is no multiple of 8: bit-identical factors
    rt = nat.get_runtime()
    pitch = 163
    mb = np.zeros((203, pitch), np.uint8)
    mb[:, :157] = mask
    dev_mask = DeviceMatrix(rt.mem.from_host(mb), (203, 157), pitch)
    plans = []
    for m in (mask, dev_mask):
        plan = DevicePlan(types, n, rank, [('a', 'b', R['a', 'b'][0], m)], [], nat.SKF_DFMC, dtype='bf16')
        for t in types:
            plan.set_factor(t, G0[t, t])
        plan.iterate(3)
        plans.append([plan.get_factor(t) for t in types])
        plan.close()
    for x, y in zip(*plans):
        np.testing.assert_array_equal(x, y)


def test_relation_sqerr_counts_the_partials_of_the_tile_it_launches():
    """Regression (round-1 advisor finding): an all-f64 residual product of a small relation with
    64 <= c_j <= 1024 runs on the deep 32 x 32 tile; the sum must cover one partial per workgroup of
    THAT tile.  100 x 140 relation, ranks 8 / 64 (and 70 / 96), all three engines."""
    from skfusion_amd._engine import DevicePlan
    rs = np.random.RandomState(31)
    for ranks in ((8, 64), (70, 96)):
        types, n, rank = ['a', 'b'], {'a': 100, 'b': 140}, {'a': ranks[0], 'b': ranks[1]}
        Rm = rs.rand(100, 140)
        G0 = {t: rs.rand(n[t], rank[t]) + 0.1 for t in types}
        for dtype, tol in ((('f64', 1e-9), ('f32', 1e-5), ('bf16', 5e-3)) if ranks[0] < 64 else (('f64', 1e-9),)):
            plan = DevicePlan(types, n, rank, [('a', 'b', Rm, None)], [], nat.SKF_DFMF, dtype=dtype)
            for t in types:
                plan.set_factor(t, G0[t])
            plan.iterate(2)
            Gd = {t: plan.get_factor(t) for t in types}
            Rref = nat.from_bf16_bits(nat.to_bf16_bits(Rm)).astype(np.float64) if dtype == 'bf16' else Rm
            want = np.linalg.norm(Rref - Gd['a'] @ plan.get_backbone(0) @ Gd['b'].T) ** 2
            got = plan.relation_sqerr(0)
            assert abs(got - want) < tol * want, (ranks, dtype, got, want)
            plan.close()


@pytest.mark.parametrize('tile', ['', '128'])
def test_fixed_staging_mode_kernels_f32_and_bf16(monkeypatch, tile):
    """(SKF_SIDE_TILE=128 keeps the 128 x 128 side-update kernels under test; default is 64 x 64.)
    Ranks above 64 and multiples of 4 with 16-byte aligned operands select the kernels whose staging
    modes are compile-time constants (f32 relation contractions P / Q, both layouts of the fused side
    update): f32 and bf16 engines against the oracle, object counts that leave row / K tails."""
    if tile:
        monkeypatch.setenv('SKF_SIDE_TILE', tile)
    rs = np.random.RandomState(21)
    types, n, rank = ['a', 'b'], {'a': 132, 'b': 148}, {'a': 68, 'b': 72}
    R = {('a', 'b'): [rs.rand(132, 148)]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.1 for t in types}
    Go, So = orc.dfmf(R, {}, types, rank, max_iter=2, G0=G0)
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=2, G0=G0, dtype='f32')
    for t in types:
        assert relerr(G[t, t], Go[t, t]) < 1e-4
    e, eo = orc.relation_errors(R, G, S), orc.relation_errors(R, Go, So)
    assert abs(e['a', 'b'][0] - eo['a', 'b'][0]) < 1e-5 * eo['a', 'b'][0]
    Rb = {k: [nat.from_bf16_bits(nat.to_bf16_bits(v[0])).astype(np.float64)] for k, v in R.items()}
    Gob, Sob = orc.dfmf(Rb, {}, types, rank, max_iter=2, G0=G0)
    Gb, Sb = _dfmf.dfmf(R, {}, types, rank, max_iter=2, G0=G0, dtype='bf16')
    for t in types:
        assert relerr(Gb[t, t], Gob[t, t]) < 2e-2


def test_relation_sqerr_and_stopping_path():
    R, types, rank = readme_graph()
    z = golden('c1_readme_dfmf.npz')
    G0 = g0_from(z, 'random/', types)
    seen = []
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=5, G0=G0, compute_err=True,
                      stopping=(('t1', 't2'), 1e-9), callback=lambda g, s, it: seen.append(it))
    assert seen == list(range(5))
    from skfusion_amd._engine import DevicePlan, flatten_relations
    rel = flatten_relations(R)
    plan = DevicePlan(types, {'t1': 50, 't2': 100, 't3': 40}, rank, rel, [], nat.SKF_DFMF)
    for t in types:
        plan.set_factor(t, G0[t, t])
    plan.iterate(3)
    Gd = {(t, t): plan.get_factor(t) for t in types}
    for k, (i, j, Rm, _) in enumerate(rel):
        Sd = plan.get_backbone(k)
        want = np.linalg.norm(Rm - Gd[i, i] @ Sd @ Gd[j, j].T) ** 2
        assert abs(plan.relation_sqerr(k) - want) < 1e-9 * want
    plan.close()


def test_shape_mismatch_is_a_hard_error():
    from skfusion_amd.fusion import DataFusionError
    R = {('a', 'b'): [np.ones((4, 5))], ('a', 'c'): [np.ones((3, 2))]}
    with pytest.raises(DataFusionError):
        _dfmf.dfmf(R, {}, ['a', 'b', 'c'], {'a': 2, 'b': 2, 'c': 2}, max_iter=1,
                   random_state=np.random.RandomState(0))


def test_transform_honours_stopping_system_and_compute_err():
    """Fold-in with `stopping_system` (reference _dfmf.py:367-376, 433-450): the system error of the new relations is
    evaluated every iteration and the loop stops on its change -- the round-1 engine dropped these options silently."""
    z = golden('transform_readme.npz')
    G = {(t, t): z['G_%s' % t] for t in TYPES}
    S = {('t1', 't2'): [z['S_t1_t2']], ('t1', 't3'): [z['S_t1_t3']], ('t2', 't1'): [z['S_t2_t1']]}
    Rn = {('t1', 't2'): [z['new_t1_t2']], ('t1', 't3'): [z['new_t1_t3']], ('t2', 't1'): [z['new_t2_t1']]}
    rank = {'t1': 10, 't2': 20, 't3': 30}
    G0 = z['random_c/G0']
    seen = []
    Gi = _dfmf.transform(Rn, {('t1', 't1'): [z['theta_t1']]}, 't1', rank, G, S, max_iter=100, G0=G0,
                         stopping_system=0.2, callback=lambda g, it: seen.append(it))
    assert 2 < len(seen) < 100
    # the same number of plain iterations gives the same factor; the oracle agrees on the stopping iteration
    Gp = _dfmf.transform(Rn, {('t1', 't1'): [z['theta_t1']]}, 't1', rank, G, S, max_iter=len(seen), G0=G0)
    assert relerr(Gi, Gp) < 1e-12
    errs = []
    Gh = G0
    for it in range(len(seen) + 3):
        Gh = orc.transform(Rn, {('t1', 't1'): [z['theta_t1']]}, 't1', rank, G, S, max_iter=it + 1, G0=G0)
        s = 0.0
        for (i, j), mats in Rn.items():
            Gi_, Gj_ = (Gh if i == 't1' else G[i, i]), (Gh if j == 't1' else G[j, j])
            s += np.linalg.norm(mats[0] - Gi_ @ S[i, j][0] @ Gj_.T)
        errs.append(s)
    stop_at = next(it for it in range(2, len(errs)) if errs[it - 2] - errs[it - 1] < 0.2)
    assert stop_at == len(seen)


def test_relation_pipelined_schedule_matches_the_staged_one_and_the_oracle(monkeypatch):
    """DFMF with every rank above 64 runs the relation-pipelined schedule (contractions on the main stream, each
    relation's backbone / B terms / side products on the second stream underneath the next relation's
    contractions, relations walked most-expensive first).  Same arithmetic as the staged schedule
    (SKF_NO_PIPELINE=1): f64 agrees with it and with the oracle to 1e-10, bf16 to its engine tolerance (the GPU suite
    runs the scaled config 3 through the pipeline in all three engines)."""
    rs = np.random.RandomState(5)
    types = ['a', 'b', 'c']
    n = {'a': 100, 'b': 120, 'c': 90}
    rank = {'a': 66, 'b': 72, 'c': 68}
    R = {('a', 'b'): [rs.rand(100, 120)], ('a', 'c'): [rs.rand(100, 90) - 0.3], ('b', 'c'): [rs.rand(120, 90)],
         ('c', 'a'): [rs.rand(90, 100)]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    Go, So = orc.dfmf(R, {}, types, rank, max_iter=2, G0=G0)
    out = {}
    for mode, dtypes in (('pipelined', ('f64', 'bf16')), ('staged', ('f64',))):
        if mode == 'staged':
            monkeypatch.setenv('SKF_NO_PIPELINE', '1')
        for dtype in dtypes:
            out[mode, dtype] = _dfmf.dfmf(R, {}, types, rank, max_iter=2, G0=G0, dtype=dtype)
    for t in types:
        assert relerr(out['pipelined', 'f64'][0][t, t], Go[t, t]) < 1e-10
        assert relerr(out['pipelined', 'f64'][0][t, t], out['staged', 'f64'][0][t, t]) < 1e-11   # (summation order)
        assert relerr(out['pipelined', 'bf16'][0][t, t], Go[t, t]) < 2e-2
    for k in So:
        assert relerr(out['pipelined', 'f64'][1][k][0], So[k][0]) < 1e-10


def test_fit_with_a_rank_above_256(monkeypatch):
    """A rank above 256 inside a fit (round 5): the pseudo-inverses of one pass are ONE batch of step-per-launch sweeps --
    orders 300, 70 and 40 side by side through sweep_step_kernel<true> -- where the blocked Cholesky inverse + unpack ran
    before (SKF_SWEEP_BIG=0).  Both against the oracle, f64; the relation of the small type is walked by the pipeline's
    mixed-rank rules (staged schedule here: one rank is below 65)."""
    rs = np.random.RandomState(17)
    types = ['a', 'b', 'c']
    n = {'a': 420, 'b': 150, 'c': 90}
    rank = {'a': 300, 'b': 70, 'c': 40}
    R = {('a', 'b'): [rs.rand(420, 150)], ('a', 'c'): [rs.rand(420, 90) - 0.3], ('b', 'c'): [rs.rand(150, 90)]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    Go, So = orc.dfmf(R, {}, types, rank, max_iter=2, G0=G0)
    for big in ('1', '0'):
        monkeypatch.setenv('SKF_SWEEP_BIG', big)
        G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=2, G0=G0, dtype='f64')
        for t in types:
            assert relerr(G[t, t], Go[t, t]) < 1e-11, (big, t)          # measured 1e-13 .. 1e-12 (either route)
        for k in So:
            assert relerr(S[k][0], So[k][0]) < 8e-11, (big, k)        # measured 2e-12 .. 8e-12


def test_fit_with_a_rank_deficient_gram_above_order_256():
    """Round 6 (reference tests/test_n_run.py:14 at a rank above 256: more latent dimensions than objects).  Type `a` has
    150 objects and rank 280: its Gram matrix has rank 150, the fast path declines it every iteration and the deflation
    over several workgroups (pchol_step_kernel, gated finishing products) forms the pseudo-inverse -- inside the batch that
    also holds a full-rank order-70 matrix, which the same launches must leave alone.  Against the oracle (scipy pinv), f64."""
    rs = np.random.RandomState(31)
    types = ['a', 'b', 'c']
    n = {'a': 150, 'b': 220, 'c': 90}
    rank = {'a': 280, 'b': 70, 'c': 40}
    R = {('a', 'b'): [rs.rand(150, 220)], ('a', 'c'): [rs.rand(150, 90) - 0.3], ('b', 'c'): [rs.rand(220, 90)]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    Go, So = orc.dfmf(R, {}, types, rank, max_iter=2, G0=G0)
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=2, G0=G0, dtype='f64')
    for t in types:
        assert relerr(G[t, t], Go[t, t]) < 1e-9, t
    for k in So:
        assert relerr(S[k][0], So[k][0]) < 1e-8, k


@pytest.mark.parametrize('dtype', ['f64', 'bf16'])
def test_round5_schedule_switches_keep_every_bit(dtype, monkeypatch):
    """Round 5: (a) split-K Gram products compute only the tiles on / below the diagonal and the reduce mirrors the rest
    (SKF_GRAM_SYM=0: every tile) -- element (a, b) and (b, a) are the same products added in the same order; (b) a type
    whose last relation is through is updated on the second stream at once (SKF_EARLY_UPDATE=0: at the end of the
    iteration); (c) the independent c x c products of a relation's chain go out two to a launch (SKF_CHAIN_PAIRS=0: one
    by one).  None changes any arithmetic: factors and backbones are bit for bit those of the plain schedule.
    Ranks 66 / 192 / 68 over 512+ objects: the Gram products are split over K and the order-192 one skips a tile; type
    `b` finishes with the first relation pair and is updated early."""
    rs = np.random.RandomState(9)
    types = ['a', 'b', 'c']
    n = {'a': 528, 'b': 544, 'c': 512}
    rank = {'a': 66, 'b': 192, 'c': 68}
    R = {('a', 'b'): [rs.rand(528, 544)], ('b', 'a'): [rs.rand(544, 528)], ('a', 'c'): [rs.rand(528, 512) - 0.2]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    runs = {}
    for name, env in (('plain', {'SKF_GRAM_SYM': '0', 'SKF_EARLY_UPDATE': '0'}), ('sym', {'SKF_EARLY_UPDATE': '0'}),
                      ('early', {'SKF_GRAM_SYM': '0'}), ('unpaired', {'SKF_GRAM_SYM': '0', 'SKF_EARLY_UPDATE': '0', 'SKF_CHAIN_PAIRS': '0'})):
        for k in ('SKF_GRAM_SYM', 'SKF_EARLY_UPDATE', 'SKF_CHAIN_PAIRS'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        runs[name] = _dfmf.dfmf(R, {}, types, rank, max_iter=2, G0=G0, dtype=dtype)
    Gp, Sp = runs['plain']
    for name in ('sym', 'early', 'unpaired'):
        G, S = runs[name]
        for t in types:
            assert np.array_equal(G[t, t], Gp[t, t]), (name, t)
        for k in Sp:
            assert np.array_equal(S[k][0], Sp[k][0]), (name, k)
    if dtype == 'f64':
        Go, So = orc.dfmf(R, {}, types, rank, max_iter=2, G0=G0)
        for t in types:
            assert relerr(Gp[t, t], Go[t, t]) < 1e-10


@pytest.mark.parametrize('dtype', ['f32'])
def test_gram_products_of_all_types_in_one_launch_keep_every_bit(dtype, monkeypatch, three=False):
    """Round 6: the Gram matrices of all types leave ONE grouped product launch and ONE reduce launch
    (gemm_mfma_group_kernel / splitk_reduce_z16_group_kernel; SKF_GRAM_GROUP=0: a product and a reduce launch per type).
    Tile list, K slices and summation order of every product are unchanged: factors and backbones bit for bit, two launches
    fewer per iteration and type beyond the first (two types and the f32 engine here; three types and every engine on the hardware).
    1 040+ objects: the symmetric split takes 8 slices, the threshold of the group."""
    from skfusion_amd._engine import launch_count
    rs = np.random.RandomState(19)
    types = ['a', 'b', 'c'] if three else ['a', 'b']
    n = {'a': 1100, 'b': 1056, 'c': 1040}
    rank = {'a': 68, 'b': 132 if three else 72, 'c': 72}     # (multiples of 4: the f32 factors then stage as vectors on the big tile)
    R = {('a', 'b'): [rs.rand(1100, 1056)]}
    if three:
        R['a', 'c'] = [rs.rand(1100, 1040) - 0.2]
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    runs, launches = {}, {}
    for name, env in (('one by one', {'SKF_GRAM_GROUP': '0'}), ('grouped', {})):
        monkeypatch.delenv('SKF_GRAM_GROUP', raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        before = launch_count()
        runs[name] = _dfmf.dfmf(R, {}, types, rank, max_iter=2, G0=G0, dtype=dtype)
        launches[name] = launch_count() - before
    assert launches['one by one'] - launches['grouped'] == 2 * 2 * (len(types) - 1), launches       # (two iterations)
    Gp, Sp = runs['one by one']
    G, S = runs['grouped']
    for t in types:
        assert np.array_equal(G[t, t], Gp[t, t]), t
    for k in Sp:
        assert np.array_equal(S[k][0], Sp[k][0]), k
    if dtype == 'f64':
        Go, So = orc.dfmf(R, {}, types, rank, max_iter=2, G0=G0)
        for t in types:
            assert relerr(G[t, t], Go[t, t]) < 1e-10


def test_dfmc_runs_the_relation_pipeline_with_the_completion_between_its_contractions(monkeypatch):
    """DFMC with a rank above 64 runs the relation pipeline too: a masked relation is contracted once before its
    completion (through the narrower factor) and twice after it; its backbone and reconstruction operands are computed
    on the second stream while the unmasked relations are contracted.  Mixed ranks (one below 64), a masked relation, a
    relation with a None mask, a sparse constraint: same iterates as the staged schedule (SKF_NO_PIPELINE=1) and as the
    oracle, f64; bf16 within its engine tolerance."""
    rs = np.random.RandomState(9)
    types = ['u', 'm', 'g']
    n = {'u': 110, 'm': 90, 'g': 40}
    rank = {'u': 66, 'm': 70, 'g': 12}
    Rum = rs.rand(110, 90)
    Rmg = (rs.rand(90, 40) < 0.2).astype(np.float64)
    Rug = rs.rand(110, 40)
    Mum = rs.rand(110, 90) < 0.6                                   # True = unknown
    R = {('u', 'm'): [Rum], ('m', 'g'): [Rmg], ('u', 'g'): [Rug]}
    M = {('u', 'm'): [Mum], ('m', 'g'): [None], ('u', 'g'): [None]}
    Tm = 0.05 * np.eye(90)
    Tm[3, 7] = Tm[7, 3] = -0.01
    Theta = {('m', 'm'): [Tm]}
    G0 = {(t, t): rs.rand(n[t], rank[t]) + 0.05 for t in types}
    Go, So = orc.dfmc(R, M, Theta, types, rank, max_iter=3, G0=G0)
    out = {}
    for mode, dtypes in (('pipelined', ('f64', 'bf16')), ('staged', ('f64',))):
        if mode == 'staged':
            monkeypatch.setenv('SKF_NO_PIPELINE', '1')
        for dtype in dtypes:
            out[mode, dtype] = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=3, G0=G0, dtype=dtype)
    for t in types:
        assert relerr(out['pipelined', 'f64'][0][t, t], Go[t, t]) < 1e-10
        assert relerr(out['pipelined', 'f64'][0][t, t], out['staged', 'f64'][0][t, t]) < 1e-11    # (summation order)
        assert relerr(out['pipelined', 'bf16'][0][t, t], Go[t, t]) < 3e-2
    for k in So:
        assert relerr(out['pipelined', 'f64'][1][k][0], So[k][0]) < 1e-10
    np.testing.assert_array_equal(R['u', 'm'][0], Rum)              # inputs untouched (reference test_dfmc.py:62,85)


def test_binary_relations_as_bitmaps_give_the_dense_results_bit_for_bit():
    """SKF_BF16: a 0 / 1 relation travels as a bitmap (SKF_REL_BINARY, detected on the host) and is expanded to bf16
    0 / 1 in LDS -- the same operands reach the matrix cores, so factors, backbones and the residual equal those of
    the dense bf16 path exactly; DFMF with a second, real-valued relation, ranks that use both kernel widths."""
    from skfusion_amd._engine import DevicePlan, DeviceMatrix
    rs = np.random.RandomState(17)
    types, n, rank = ['m', 'a', 'u'], {'m': 150, 'a': 200, 'u': 90}, {'m': 12, 'a': 9, 'u': 7}
    Rma = (rs.rand(150, 200) < 0.05).astype(np.float64)          # movie x actor, binary, 5 % dense
    Rum = rs.rand(90, 150)                                        # user x movie, real valued
    Rua = (rs.rand(90, 200) < 0.3).astype(np.float64)
    G0 = {t: rs.rand(n[t], rank[t]) + 0.05 for t in types}
    rt = nat.get_runtime()
    out = {}
    for mode in ('bitmap', 'dense'):
        rels = []
        for i, j, M in (('m', 'a', Rma), ('u', 'm', Rum), ('u', 'a', Rua)):
            dm = DeviceMatrix(rt.mem.from_host(nat.to_bf16_bits(M.astype(np.float32))), M.shape,
                              binary=(mode == 'bitmap' and M is not Rum))
            rels.append((i, j, dm, None))
        plan = DevicePlan(types, n, rank, rels, [], nat.SKF_DFMF, dtype='bf16')
        for t in types:
            plan.set_factor(t, G0[t])
        plan.iterate(4)
        out[mode] = ([plan.get_factor(t) for t in types], [plan.get_backbone(k) for k in range(3)],
                     [plan.relation_sqerr(k) for k in range(3)])
        plan.close()
    for a, b in zip(out['bitmap'][0] + out['bitmap'][1], out['dense'][0] + out['dense'][1]):
        np.testing.assert_array_equal(a, b)
    assert out['bitmap'][2] == out['dense'][2]
    # the host layer detects 0 / 1 relations by itself, and the result is that of the f64 oracle within bf16 tolerance
    R = {('m', 'a'): [Rma], ('u', 'm'): [Rum], ('u', 'a'): [Rua]}
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=4, G0={(t, t): G0[t] for t in types}, dtype='bf16')
    for k, t in enumerate(types):
        np.testing.assert_array_equal(G[t, t], out['bitmap'][0][k])
    # a relation flagged binary that is not: refused at bind time
    bad = DeviceMatrix(rt.mem.from_host(nat.to_bf16_bits(Rum.astype(np.float32))), Rum.shape, binary=True)
    with pytest.raises(nat.SkfNativeError):
        DevicePlan(types, n, rank, [('u', 'm', bad, None), ('m', 'a', Rma, None)], [], nat.SKF_DFMF, dtype='bf16')


@pytest.mark.parametrize('dtype,wide', [('f64', True), ('bf16', False)])       # (the GPU suite adds f64 on the staged schedule)
def test_sparse_constraints_as_csr_give_the_dense_product(dtype, wide):
    """A constraint with few non-zeros (skf_theta_desc.nnz) is compacted to CSR on the device and applied as a
    row-gather product with the same +- split as the dense form (reference _dfmf.py:276-283): same factors as the
    dense product to summation order, on the staged schedule and (every rank > 64, DFMF) under the relation
    pipeline; a bound below the true count is refused at bind time."""
    from skfusion_amd._engine import DevicePlan, DeviceMatrix
    rs = np.random.RandomState(23)
    types = ['a', 'b']
    n = {'a': 130, 'b': 110}
    rank = {'a': 66, 'b': 70} if wide else {'a': 9, 'b': 6}
    R = rs.rand(130, 110)
    T1 = np.where(rs.rand(130, 130) < 0.03, rs.randn(130, 130), 0.0)       # must-link (-) and cannot-link (+)
    T1 = T1 + T1.T
    T2 = 0.5 * np.eye(130)
    T3 = np.where(rs.rand(110, 110) < 0.02, -1.0, 0.0)
    T3[7, :] = 0.0                                                          # an empty row
    G0 = {t: rs.rand(n[t], rank[t]) + 0.05 for t in types}
    rt = nat.get_runtime()
    npd = np.float64 if dtype == 'f64' else np.float32
    out = {}
    for mode in ('sparse', 'dense'):
        thetas = []
        for t, T in (('a', T1), ('a', T2), ('b', T3)):
            dm = DeviceMatrix(rt.mem.from_host(np.ascontiguousarray(T, dtype=npd)), T.shape)
            dm.nnz = int(np.count_nonzero(T)) if mode == 'sparse' else 0
            thetas.append((t, dm))
        plan = DevicePlan(types, n, rank, [('a', 'b', R, None)], thetas, nat.SKF_DFMF, dtype=dtype)
        for t in types:
            plan.set_factor(t, G0[t])
        plan.iterate(3)
        out[mode] = [plan.get_factor(t) for t in types]
        plan.close()
    # SKF_BF16: the dense form rounds Theta and G to bf16 for the matrix cores, the CSR form works on the f32 masters
    tol = 1e-12 if dtype == 'f64' else 5e-3
    for a, b in zip(out['sparse'], out['dense']):
        assert relerr(a, b) < tol
    Go, _ = orc.dfmf({('a', 'b'): [R]}, {('a', 'a'): [T1, T2], ('b', 'b'): [T3]}, types, rank, max_iter=3,
                     G0={(t, t): G0[t] for t in types})
    for k, t in enumerate(types):
        assert relerr(out['sparse'][k], Go[t, t]) < (1e-10 if dtype == 'f64' else 2e-2)
    low = DeviceMatrix(rt.mem.from_host(np.ascontiguousarray(T1, dtype=npd)), T1.shape)
    low.nnz = int(np.count_nonzero(T1)) - 1
    with pytest.raises(nat.SkfNativeError):
        DevicePlan(types, n, rank, [('a', 'b', R, None)], [('a', low)], nat.SKF_DFMF, dtype=dtype)


def test_very_sparse_binary_relation_is_contracted_by_row_gathers():
    """SKF_BF16: a 0 / 1 relation with at most 1 entry in 256 set keeps the positions of its ones as CSR and CSC (built
    from the bitmap at bind time, ascending) and P = R G_j, Q = R^T G_i become gathers of the f32 factor rows -- exact f32
    sums, where the bitmap path rounds G to bf16 first: P and Q equal the host products to f32 rounding (the bitmap path
    would be off by 1e-3), rows and columns without ones give zeros, and a fit agrees with the oracle."""
    from skfusion_amd._engine import DevicePlan, DeviceMatrix
    rs = np.random.RandomState(31)
    types, n, rank = ['m', 'a'], {'m': 210, 'a': 300}, {'m': 8, 'a': 12}
    A = (rs.rand(210, 300) < 0.002).astype(np.float64)
    A[5, :] = 0.0
    A[:, 17] = 0.0
    A[100, 299] = 1.0
    A[209, 0] = 1.0
    A[:, 123] = 0.0
    A[::3, 123] = 1.0                     # a heavy column (70 ones): the long-column branch of the CSC build
    assert 0 < A.sum() <= 210 * 300 // 256
    G0 = {t: (rs.rand(n[t], rank[t]) + 0.05).astype(np.float32) for t in types}
    rt = nat.get_runtime()
    dm = DeviceMatrix(rt.mem.from_host(nat.to_bf16_bits(A.astype(np.float32))), A.shape, binary=True)
    plan = DevicePlan(types, n, rank, [('m', 'a', dm, None)], [], nat.SKF_DFMF, dtype='bf16')
    for t in types:
        plan.set_factor(t, G0[t])
    plan.iterate(1)
    P, Q = plan.get_contraction(0, 0), plan.get_contraction(0, 1)
    plan.close()
    assert relerr(P, A @ G0['a'].astype(np.float64)) < 1e-6
    assert relerr(Q, A.T @ G0['m'].astype(np.float64)) < 1e-6
    assert not P[5].any() and not Q[17].any()
    # a denser relation of the same shape stays on the bitmap kernel (bf16-rounded factor: 1e-3 off the exact product)
    B = (rs.rand(210, 300) < 0.05).astype(np.float64)
    dmb = DeviceMatrix(rt.mem.from_host(nat.to_bf16_bits(B.astype(np.float32))), B.shape, binary=True)
    plan = DevicePlan(types, n, rank, [('m', 'a', dmb, None)], [], nat.SKF_DFMF, dtype='bf16')
    for t in types:
        plan.set_factor(t, G0[t])
    plan.iterate(1)
    Pb = plan.get_contraction(0, 0)
    plan.close()
    assert 1e-5 < relerr(Pb, B @ G0['a'].astype(np.float64)) < 1e-2
    # through the host layer (0 / 1 detection), three iterations against the oracle
    R = {('m', 'a'): [A]}
    G0d = {(t, t): G0[t].astype(np.float64) for t in types}
    G, S = _dfmf.dfmf(R, {}, types, rank, max_iter=3, G0=G0d, dtype='bf16')
    Go, So = orc.dfmf(R, {}, types, rank, max_iter=3, G0=G0d)
    for t in types:
        assert relerr(G[t, t], Go[t, t]) < 1e-4


def sparse_binary_lists_case(n, rank, density, seed, tol=2e-6):
    """SKF_BF16, ranks 64 / 128 / 256: a 0 / 1 relation with at most 1 entry in 80 set is contracted as LISTS over the bf16
    rows of the factors (srp_bf16_v6_kernel<.., SRP_ONES>: 8-byte chunks at rank 64, one / two 16-byte chunks per lane at
    128 / 256; lists in parts where the gathered factor is large) -- P = R G_j and Q = R^T G_i are f32 sums of bf16-rounded
    factor rows: equal to the host product of the ROUNDED factors to f32 rounding (what the bitmap kernels compute too)."""
    from skfusion_amd._engine import DevicePlan, DeviceMatrix
    rs = np.random.RandomState(seed)
    types = ['m', 'a', 'c']
    A = (rs.rand(n['m'], n['a']) < density).astype(np.float64)
    A[5, :] = 0.0
    A[:, 17] = 0.0
    A[::3, 23] = 1.0                                                  # a heavy column
    A[n['m'] - 1, n['a'] - 1] = 1.0
    C = (rs.rand(n['c'], n['m']) < density).astype(np.float64)
    C[0, 0] = 1.0
    assert A.sum() <= A.size // 80 and C.sum() <= C.size // 80
    G0 = {t: (rs.rand(n[t], rank[t]) + 0.05).astype(np.float32) for t in types}
    Gb = {t: nat.from_bf16_bits(nat.to_bf16_bits(G0[t])).astype(np.float64) for t in types}     # what the lists gather
    rt = nat.get_runtime()
    dev = lambda M: DeviceMatrix(rt.mem.from_host(nat.to_bf16_bits(M.astype(np.float32))), M.shape, binary=True)
    plan = DevicePlan(types, n, rank, [('m', 'a', dev(A), None), ('c', 'm', dev(C), None)], [], nat.SKF_DFMF, dtype='bf16')
    for t in types:
        plan.set_factor(t, G0[t])
    plan.iterate(1)
    got = [plan.get_contraction(0, 0), plan.get_contraction(0, 1), plan.get_contraction(1, 0), plan.get_contraction(1, 1)]
    plan.close()
    want = [A @ Gb['a'], A.T @ Gb['m'], C @ Gb['m'], C.T @ Gb['c']]
    worst = max(relerr(g, w) for g, w in zip(got, want))
    assert worst < tol, worst
    assert not got[0][5].any() and not got[1][17].any()
    # the fit itself: the same graph with the relations as bitmaps only differs by the order of the f32 sums
    return worst


@pytest.mark.parametrize('parts', [0, 4])
def test_sparse_binary_relations_as_lists_over_bf16_rows(parts, monkeypatch):
    """parts = 4: SKF_KNOWN_PARTS cuts the (short) lists into four parts of the inner index -- parted_ptr_kernel's binary
    search, empty segments, partial outputs summed by sum_parts_kernel; 0: the engine's choice (one part at this size)."""
    if parts:
        monkeypatch.setenv('SKF_KNOWN_PARTS', str(parts))
    sparse_binary_lists_case({'m': 96, 'a': 1500, 'c': 300}, {'m': 64, 'a': 128, 'c': 256}, 0.01, 33 + parts)


def test_fold_in_of_binary_new_relations_bf16():
    """SKF_TRANSFORM with 0 / 1 new-object relations in the bf16 engine: one kept as a bitmap, one sparse enough for the
    CSR / CSC gathers, in both orientations (target on the row and on the column side) -- against the oracle's
    transform on the same frozen model."""
    rs = np.random.RandomState(41)
    types, rank = ['t', 'a', 'b'], {'t': 7, 'a': 9, 'b': 6}
    na, nb, nt = 220, 260, 40
    Gf = {('a', 'a'): rs.rand(na, 9) + 0.05, ('b', 'b'): rs.rand(nb, 6) + 0.05, ('t', 't'): rs.rand(30, 7)}
    S = {('t', 'a'): [rs.rand(7, 9)], ('b', 't'): [rs.rand(6, 7)]}
    Rta = (rs.rand(nt, na) < 0.3).astype(np.float64)                  # bitmap
    Rbt = (rs.rand(nb, nt) < 0.003).astype(np.float64)                # 1 in 256 at most: gathers
    Rbt[7, 3] = 1.0
    assert 0 < Rbt.sum() <= nb * nt // 256
    Rn = {('t', 'a'): [Rta], ('b', 't'): [Rbt]}
    G0 = rs.rand(nt, 7) + 0.05
    want = orc.transform(Rn, {}, 't', rank, Gf, S, max_iter=20, G0=G0)
    got64 = _dfmf.transform(Rn, {}, 't', rank, Gf, S, max_iter=20, G0=G0)
    assert relerr(got64, want) < 1e-10
    got = _dfmf.transform(Rn, {}, 't', rank, Gf, S, max_iter=20, G0=G0, dtype='bf16')
    assert relerr(got, want) < 1e-2


@pytest.mark.parametrize('dtype,parts,rank_a', [('f64', 1, 64), ('f64', 4, 64), ('f32', 1, 64), ('f32', 1, 20), ('bf16', 2, 64),
                                               ('bf16', 2, 128), ('bf16', 1, 256)])
def test_dfmc_on_the_known_entries_only_matches_the_dense_completion(dtype, parts, rank_a, monkeypatch):
    """skf_relation_desc.known_bound: a masked relation kept as lists of its known entries (csrc/skf_known.h) -- the
    completed relation of _dfmc.py:319-325 is never formed.  f64: equal to the dense path to rounding; rank 64 on the row
    type takes the 16-byte-chunk list kernels (f64: 32 lanes per vector, f32: 16, bf16: 8 with v_dot2 + DPP), rank 20 the
    any-width kernel; bf16 at rank 128 / 256: srp_bf16_v6_kernel (one / two chunks per lane, zero row for the list tails;
    at 256 the residual passes stay on the two-rows-per-vector kernel).  (bf16: the dense path rounds every completed entry to bf16, the lists do not.)"""
    import known_cases as K
    n, ranks = {'a': 150, 'b': 130, 'c': 40}, {'a': rank_a, 'b': 24, 'c': 5}
    # (G, S, squared errors, P S^T, Q); measured f64 <= 3e-14 / 2e-13 / 1.2e-14 / 1.3e-13 / 1.3e-13, bf16 <= 1.4e-3 / 2e-2 / 1e-3 / 3e-2 / 2e-3
    tol = {'f64': (2e-13, 1e-12, 1e-13, 1e-12, 1e-12), 'f32': (2e-6, 1e-5, 1e-6, 1e-5, 2e-6),
           'bf16': (7e-3, 1e-1, 5e-3, 1.5e-1, 1e-2)}[dtype]
    K.sparse_against_dense(n, ranks, 0.06, 3, dtype, tol, 'emulator %s parts %d rank %d' % (dtype, parts, rank_a), monkeypatch, parts)


def test_known_entries_dfmc_c5_golden_and_bound_too_small(monkeypatch):
    """The scaled config 5 (2 % of the ratings known) takes the list path by default and reproduces the reference golden;
    a bound below the true count is refused at bind time."""
    from helpers import movielens_style_graph
    from skfusion_amd._engine import DevicePlan, flatten_relations, flatten_thetas, count_objects, pack_mask
    z = golden('c5_movielens_scaled.npz')
    R, M, Theta, types, rank = movielens_style_graph()
    snaps = Snapshots(range(6))
    _dfmc.dfmc(R, M, Theta, types, rank, max_iter=6, callback=snaps, G0=g0_from(z, 'dfmc/', types))
    assert compare_snapshots(z, 'dfmc/', snaps.snap, 1e-10) < 1e-10
    rel = flatten_relations(R, M)
    n = count_objects(types, R)
    plan = DevicePlan(types, n, rank, rel, flatten_thetas(Theta), nat.SKF_DFMC)
    assert plan.get_contraction.__self__ is plan
    for t in types:
        plan.set_factor(t, z['dfmc/G0_%s' % t])
    plan.iterate(1)
    assert plan.get_contraction(0, 2).shape == (n['user'], rank['user'])       # the list path is the one that ran
    with pytest.raises(nat.SkfNativeError):
        plan.get_contraction(0, 0)                                             # ... and it never forms P
    plan.close()
    import emul.runtime as er
    mem = nat._runtime.mem
    pm = pack_mask(rel[0][3], mem)
    pm.known = pm.known - 5
    with pytest.raises(nat.SkfNativeError, match='known entries'):
        DevicePlan(types, n, rank, [(rel[0][0], rel[0][1], rel[0][2], pm)] + rel[1:], flatten_thetas(Theta), nat.SKF_DFMC)
