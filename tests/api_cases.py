"""Class-layer scenarios re-stating the assertions of the reference's own tests
(skfusion/tests/test_dfmf.py, test_dfmc.py, test_n_run.py, test_multiple_relations.py,
test_base.py).  `full=True` (GPU) runs the reference's iteration counts and exact-reconstruction
asserts; `full=False` (host emulator) runs a few iterations and checks shapes / consistency."""
import numpy as np

from skfusion_amd.fusion import (Relation, ObjectType, FusionGraph, Dfmf, Dfmc, DfmfTransform,
                                 DataFusionError)


def iters(full, n=100):
    return n if full else 3


def exact_reconstruction(cls, full):
    """test_dfmf.py:9-23 / test_dfmc.py:9-23: full-rank factorisation reproduces the data."""
    rnds = np.random.RandomState(0)
    R12 = rnds.rand(50, 30)
    t1, t2 = ObjectType('type1', 50), ObjectType('type2', 30)
    relation = Relation(R12, t1, t2)
    graph = FusionGraph()
    graph.add_relation(relation)
    fuser = cls(init_type='random', random_state=rnds, max_iter=iters(full)).fuse(graph)
    assert fuser.backbone(relation).shape == (50, 30)
    assert fuser.factor(t1).shape == (50, 50) and fuser.factor(t2).shape == (30, 30)
    if full:
        np.testing.assert_almost_equal(fuser.complete(relation), relation.data)
    np.testing.assert_array_equal(relation.data, R12)


def non_finite_inputs(full):
    """test_dfmf.py:25-46: NaN / inf / masked inputs -> all-finite completion."""
    rnds = np.random.RandomState(0)
    R12, R13 = rnds.rand(50, 30), rnds.rand(50, 10)
    R12 = np.ma.masked_greater(R12, 0.7)
    R12[R12 < 0.1] = np.nan
    R13[R13 < 0.5] = np.inf
    t1, t2, t3 = ObjectType('type1', 50), ObjectType('type2', 30), ObjectType('type3', 10)
    relations = [Relation(R12, t1, t2, fill_value='row_mean'),
                 Relation(R13, t1, t3, fill_value='col_mean')]
    fuser = Dfmf(init_type='random', random_state=rnds, max_iter=iters(full)).fuse(FusionGraph(relations))
    assert fuser.backbone(relations[0]).shape == (50, 30)
    assert fuser.backbone(relations[1]).shape == (50, 10)
    assert np.sum(np.isfinite(fuser.complete(relations[0]))) == R12.size


def masked_completion(full):
    """test_dfmc.py:25-39: the reconstruction equals the data on the unmasked entries."""
    rnds = np.random.RandomState(0)
    R12 = np.ma.masked_greater(rnds.rand(50, 30), 0.7)
    t1, t2 = ObjectType('type1', 50), ObjectType('type2', 30)
    relation = Relation(R12, t1, t2)
    fuser = Dfmc(init_type='random', random_state=rnds, max_iter=iters(full)).fuse(FusionGraph([relation]))
    assert fuser.backbone(relation).shape == (50, 30)
    if full:
        R12_hat = fuser.complete(relation)
        np.testing.assert_almost_equal(R12_hat[~R12.mask], R12.data[~R12.mask])


def processors(cls, full):
    """test_dfmf.py:80-120 / test_dfmc.py:41-85: pre/post-processor hooks, data untouched."""
    rnds = np.random.RandomState(0)
    R12 = rnds.rand(50, 30)
    keep = R12.copy()
    t1, t2 = ObjectType('type1', 50), ObjectType('type2', 30)
    rel = Relation(R12, t1, t2, preprocessor=lambda d: np.ones_like(d))
    fuser = cls(init_type='random', random_state=rnds, max_iter=iters(full)).fuse(FusionGraph([rel]))
    if full:
        np.testing.assert_almost_equal(fuser.complete(rel), np.ones_like(R12))
    np.testing.assert_array_equal(rel.data, keep)
    rel = Relation(R12, t1, t2, postprocessor=lambda d: d - np.mean(d))
    fuser = cls(init_type='random', random_state=rnds, max_iter=iters(full)).fuse(FusionGraph([rel]))
    if full:
        np.testing.assert_almost_equal(fuser.complete(rel), R12 - np.mean(R12))
    np.testing.assert_array_equal(rel.data, keep)


def several_runs(cls, full):
    """test_n_run.py: n_run=3 -> generators of length 3; rank 50 > 30 objects (pinv truncation)."""
    rnds = np.random.RandomState(0)
    R12, R13 = rnds.rand(30, 30), rnds.rand(30, 30)
    t1, t2, t3 = ObjectType('type1', 50), ObjectType('type2', 30), ObjectType('type3', 10)
    relations = [Relation(R12, t1, t2), Relation(R13, t1, t3)]
    graph = FusionGraph()
    graph.add_relations_from(relations)
    fuser = cls(init_type='random', random_state=rnds, n_run=3, max_iter=iters(full)).fuse(graph)
    for ot in (t1, t2, t3):
        factors = list(fuser.factor(ot))
        assert len(factors) == 3
        for f in factors:
            assert f.shape == (30, ot.rank) and np.isfinite(f).all()
    assert len(list(fuser.backbone(relations[0]))) == 3
    assert len(list(fuser.backbone(relations[1]))) == 3
    assert len(list(fuser.complete(relations[1]))) == 3
    G1, S13, G3 = fuser.factor(t1, run=1), fuser.backbone(relations[1], run=1), fuser.factor(t3, run=1)
    np.testing.assert_almost_equal(fuser.complete(relations[1], run=1), G1.dot(S13).dot(G3.T))
    # the three restarts differ (one shared RandomState, consumed sequentially)
    assert not np.allclose(fuser.factor(t1, run=0), fuser.factor(t1, run=2))
    return fuser


def multiple_relations(cls, full):
    """test_multiple_relations.py: two relations between one pair -> one backbone each."""
    rnds = np.random.RandomState(0)
    R12a, R12b, R13 = rnds.rand(30, 20), rnds.rand(30, 20), rnds.rand(30, 10)
    t1, t2, t3 = ObjectType('type1', 8), ObjectType('type2', 6), ObjectType('type3', 4)
    relations = [Relation(R12a, t1, t2), Relation(R12b, t1, t2), Relation(R13, t1, t3)]
    fuser = cls(init_type='random', random_state=rnds, max_iter=iters(full, 50)).fuse(FusionGraph(relations))
    S = [fuser.backbone(r) for r in relations]
    assert S[0].shape == (8, 6) and S[1].shape == (8, 6) and S[2].shape == (8, 4)
    assert not np.allclose(S[0], S[1])
    for r, s in zip(relations, S):
        want = fuser.factor(r.row_type).dot(s).dot(fuser.factor(r.col_type).T)
        np.testing.assert_almost_equal(fuser.complete(r), want)


def pipeline_and_transform(full):
    """test_base.py:9-38 + test_dfmf.py:48-78: default init ('random_c'), fold-in of new rows."""
    rnds = np.random.RandomState(0)
    R12, R13, R23 = rnds.rand(50, 30), rnds.rand(50, 40), rnds.rand(30, 40)
    t1, t2, t3 = ObjectType('type1', 30), ObjectType('type2', 40), ObjectType('type3', 40)
    relations = [Relation(R12, t1, t2), Relation(R13, t1, t3), Relation(R23, t2, t3)]
    fuser = Dfmf(random_state=rnds, max_iter=iters(full)).fuse(FusionGraph(relations))
    assert fuser.factor(t1).shape == (50, 30) and fuser.factor(t2).shape == (30, 40)
    assert fuser.backbone(relations[2]).shape == (40, 40)
    new_graph = FusionGraph([Relation(rnds.rand(15, 30), t1, t2), Relation(rnds.rand(15, 40), t1, t3)])
    tr = DfmfTransform(random_state=rnds, max_iter=iters(full)).transform(t1, new_graph, fuser)
    assert tr.factor(t1).shape == (15, 30) and np.isfinite(tr.factor(t1)).all()
    assert [p for p in fuser.chain(t1, t3)] == [[t1, t3], [t1, t2, t3]]
    # a relation that does not touch the target is rejected (base.py:224-231)
    bad = FusionGraph([Relation(rnds.rand(30, 40), t2, t3)])
    try:
        DfmfTransform(max_iter=1).transform(t1, bad, fuser)
        raise AssertionError('expected DataFusionError')
    except DataFusionError:
        pass


def fold_in_recovers_known_rows(full):
    """test_dfmf.py:48-78: folding in two rows of the training data reproduces their factors."""
    rs = np.random.RandomState(42)
    R12 = rs.rand(5, 3)
    t1, t2 = ObjectType('type1', 2), ObjectType('type2', 2)
    relation = Relation(R12, t1, t2)
    fuser = Dfmf(init_type='random', random_state=np.random.RandomState(0), max_iter=100).fuse(
        FusionGraph([relation]))
    new_graph = FusionGraph([Relation(R12[:2].copy(), t1, t2)])
    tr = DfmfTransform(random_state=np.random.RandomState(0), max_iter=iters(full)).transform(
        t1, new_graph, fuser)
    new_G1, G1, G2, S12 = tr.factor(t1), fuser.factor(t1), fuser.factor(t2), fuser.backbone(relation)
    assert new_G1.shape == (2, 2)
    if full:
        d_hat = new_G1.dot(S12).dot(G2.T) - G1.dot(S12).dot(G2.T)[:2]
        assert np.sum(d_hat ** 2) / d_hat.size < 1e-5


def blockwise_completion():
    """complete_blocks (device, blockwise) == complete (host NumPy) for f64, close for f32."""
    rs = np.random.RandomState(2)
    t1, t2 = ObjectType('type1', 7), ObjectType('type2', 5)
    rel = Relation(rs.rand(130, 45), t1, t2)
    fuser = Dfmf(max_iter=3, init_type='random', random_state=1).fuse(FusionGraph([rel]))
    want = fuser.complete(rel)
    got = np.zeros_like(want)
    seen = []
    for sl, block in fuser.complete_blocks(rel, block_rows=50):
        got[sl] = block
        seen.append((sl.start, sl.stop))
    assert seen == [(0, 50), (50, 100), (100, 130)]
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    got32 = np.vstack([b for _, b in fuser.complete_blocks(rel, block_rows=64, dtype='f32')])
    np.testing.assert_allclose(got32, want, rtol=1e-5, atol=1e-5)
    # the backbone and the column factor are uploaded once and stay resident across the blocks; device=True
    # hands the block over without a copy back
    from skfusion_amd._engine import DeviceReconstructor
    import skfusion_amd._native as nat
    rec = DeviceReconstructor(fuser.backbone(rel), fuser.factor(t2))
    G1 = fuser.factor(t1)
    mem = nat.get_runtime().mem
    for r0 in (0, 50, 100):
        dm = rec.block(G1[r0:r0 + 50], device=True)
        blk = mem.to_host(dm.buf, dm.shape, np.float64)
        np.testing.assert_allclose(blk, want[r0:r0 + 50], rtol=1e-12, atol=1e-12)
    assert rec.uploads == 2
    dev = [mem.to_host(b.buf, b.shape, np.float64).copy() for _, b in fuser.complete_blocks(rel, block_rows=50, device=True)]
    np.testing.assert_allclose(np.vstack(dev), want, rtol=1e-12, atol=1e-12)


def device_fill_strategies():
    """Relation.filled() on the device (skf_fill_unknown) against the reference outputs pinned in
    tests/golden/fill_strategies.npz -- values to 1e-13 (the summation order of the means differs from NumPy's
    pairwise sums), infinities and the survival of the mask exactly -- and inside Dfmf / Dfmc(device_fill=True)."""
    import warnings
    import skfusion_amd._native as nat
    from helpers import golden
    z = golden('fill_strategies.npz')
    mem = nat.get_runtime().mem
    for tag in ('masked', 'plain', 'finite', 'corner', 'plaincorner', 'nomask', 'nomaskfinite', 'maskedinf'):
        if tag.startswith('plain'):
            arr = z[tag].copy()
        else:
            arr = np.ma.MaskedArray(z[tag + '_data'].copy(), mask=z[tag + '_mask'].copy())
        for fv in ('mean', 'row_mean', 'col_mean', 0.5):
            rel = Relation(arr, ObjectType('a'), ObjectType('b'), fill_value=fv)
            for dtype, npd, tol in (('f64', np.float64, 1e-13), ('f32', np.float32, 1e-6)):
                dm, mask = rel.filled_device(dtype)
                got = mem.to_host(dm.buf, dm.shape, npd).astype(np.float64)
                key = '%s/%s' % (tag, fv)
                want = z[key + '/data']
                fin = np.isfinite(want)
                np.testing.assert_array_equal(np.isfinite(got), fin)
                np.testing.assert_array_equal(got[~fin], want[~fin])              # +-inf exactly
                np.testing.assert_allclose(got[fin], want[fin], rtol=tol, atol=0)
                assert (mask is not None) == bool(z[key + '/is_masked'])
                if mask is not None:
                    np.testing.assert_array_equal(mask, z[key + '/mask'])
    # inside a fit: the device-filled graph gives the factors of the host-filled one
    rs = np.random.RandomState(8)
    t1, t2, t3 = ObjectType('type1', 4), ObjectType('type2', 3), ObjectType('type3', 2)
    A = rs.rand(30, 20)
    A[rs.rand(30, 20) < 0.1] = np.nan
    B = np.ma.masked_greater(rs.rand(30, 12), 0.8)
    for cls in (Dfmf, Dfmc):
        fits = []
        for dev in (False, True):
            g = FusionGraph([Relation(A.copy(), t1, t2, fill_value='row_mean'), Relation(B.copy(), t1, t3, fill_value=0.3)])
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                fits.append(cls(max_iter=5, init_type='random', random_state=4, device_fill=dev).fuse(g))
        for t in (t1, t2, t3):
            np.testing.assert_allclose(fits[1].factor(t), fits[0].factor(t), rtol=1e-10, atol=1e-12)


def error_paths():
    t1, t2, t9 = ObjectType('type1', 2), ObjectType('type2', 2), ObjectType('nine', 2)
    rel = Relation(np.random.RandomState(0).rand(5, 3), t1, t2)
    fuser = Dfmf(max_iter=1, init_type='random', random_state=0).fuse(FusionGraph([rel]))
    for call in (lambda: fuser.factor(t9), lambda: fuser.backbone(Relation(np.ones((5, 3)), t1, t2)),
                 lambda: fuser.complete(Relation(np.ones((5, 3)), t1, t9))):
        try:
            call()
            raise AssertionError('expected DataFusionError')
        except DataFusionError:
            pass
    assert 'Dfmf(max_iter=1' in repr(fuser)


def n_jobs_concurrent_restarts_equal_sequential(cls):
    """n_jobs > 1: the restarts run on streams of their own (one uploaded graph, one captured hipGraph per
    plan); every run must equal the one-after-the-other result bit for bit."""
    rs = np.random.RandomState(5)
    t1, t2, t3 = ObjectType('A', 6), ObjectType('B', 5), ObjectType('C', 4)
    R12 = rs.rand(40, 30)
    if cls is Dfmc:
        R12 = np.ma.masked_array(R12, mask=rs.rand(40, 30) > 0.8)
    rels = [Relation(R12, t1, t2), Relation(rs.rand(40, 20), t1, t3),
            Relation(rs.rand(30, 20), t2, t3), Relation(-0.02 * (rs.rand(30, 30) > 0.95), t2, t2)]
    graph = FusionGraph(rels)
    # n_jobs = 1: Dfmf shares every launch among the restarts of this small graph (skf_iterate_batch); compute_err needs
    # the host every iteration and so runs them strictly one after the other: the reference for both
    one = cls(max_iter=8, n_run=5, random_state=3, n_jobs=1, compute_err=True).fuse(graph)
    seq = cls(max_iter=8, n_run=5, random_state=3, n_jobs=1).fuse(graph)
    par = cls(max_iter=8, n_run=5, random_state=3, n_jobs=3).fuse(graph)
    for other in (seq, par):
        for t in (t1, t2, t3):
            for a, b in zip(one.factor(t), other.factor(t)):
                np.testing.assert_array_equal(a, b)
        for r in rels[:3]:
            for a, b in zip(one.backbone(r), other.backbone(r)):
                np.testing.assert_array_equal(a, b)


def persistence_round_trip_on_the_engine(tmpdir, dtype='f32'):
    """SURVEY.md 8 f4 on the device path: fit on the engine -> save -> load (onto the graph, and onto a skeleton graph) ->
    the loaded model drives blockwise device completion and the fold-in to the SAME bits as the model that was never
    saved (reference accessors base.py:35-56, 169-189; fold-in dfmf.py:109-115, 155-204; chaining as in
    examples/dicty_chaining.py:56-63)."""
    import os
    import skfusion_amd._native as nat
    from skfusion_amd.fusion import FusionFit
    rs = np.random.RandomState(11)
    t1, t2, t3 = ObjectType('genes', 12), ObjectType('terms', 9), ObjectType('conditions', 6)
    rels = [Relation(rs.rand(140, 90), t1, t2, name='ann'), Relation(rs.rand(140, 60), t1, t3, name='expr'),
            Relation(rs.rand(90, 60), t2, t3, name='ctx'), Relation(0.01 * np.eye(140), t1, t1, name='ppi')]
    graph = FusionGraph(rels)
    fuser = Dfmf(max_iter=8, init_type='random_vcol', random_state=5, n_run=2, dtype=dtype).fuse(graph)
    path = fuser.save(os.path.join(str(tmpdir), 'fit.npz'))
    mem = nat.get_runtime().mem
    new_graph = lambda: FusionGraph([Relation(np.random.RandomState(3).rand(20, 90), t1, t2),
                                     Relation(np.random.RandomState(4).rand(20, 60), t1, t3)])

    def fold_in(model):
        return DfmfTransform(max_iter=6, init_type='random_c', random_state=7, dtype=dtype).transform(t1, new_graph(), model)
    want_tr = fold_in(fuser).factor(t1)
    for onto in (graph, None):
        back = FusionFit.load(path, onto)
        g2 = back.fusion_graph
        typ = {ot.name: ot for ot in g2.object_types}
        rel2 = [r for r in g2.relations if r.row_type is not r.col_type]
        for run in (0, 1):
            for ot in (t1, t2, t3):
                np.testing.assert_array_equal(back.factor(typ[ot.name], run=run), fuser.factor(ot, run=run))
            for ra, rb in zip(rels[:3], rel2):
                np.testing.assert_array_equal(back.backbone(rb, run=run), fuser.backbone(ra, run=run))
                want = [mem.to_host(b.buf, b.shape, np.float64 if dtype == 'f64' else np.float32).copy()
                        for _, b in fuser.complete_blocks(ra, block_rows=64, run=run, dtype=dtype, device=True)]
                got = [mem.to_host(b.buf, b.shape, np.float64 if dtype == 'f64' else np.float32).copy()
                       for _, b in back.complete_blocks(rb, block_rows=64, run=run, dtype=dtype, device=True)]
                assert len(want) == len(got) == (ra.data.shape[0] + 63) // 64
                for w, g in zip(want, got):
                    np.testing.assert_array_equal(g, w)
        assert [[t.name for t in p] for p in back.chain(typ['genes'], typ['conditions'])] == \
               [['genes', 'conditions'], ['genes', 'terms', 'conditions']]
        if onto is not None:                     # the fold-in reads the frozen factors / backbones of the loaded model
            np.testing.assert_array_equal(fold_in(back).factor(t1), want_tr)


def probe_fusion_graph(masked):
    """The probe graph of tests/golden/probe_multirel.npz (two relations between one pair, a negative-valued relation,
    constraints on two types; `masked`: the two masks as masked arrays) as a FusionGraph.  Returns (graph, types by name)."""
    from helpers import golden, probe_graph
    R, Theta, M, types, rank = probe_graph(golden('probe_multirel.npz'))
    ot = {t: ObjectType(t, rank[t]) for t in types}
    rels = []
    for (i, j), mats in R.items():
        for l, m in enumerate(mats):
            data = np.ma.masked_array(m.copy(), mask=M[i, j][l].copy()) if (masked and M[i, j][l] is not None) else m.copy()
            rels.append(Relation(data, ot[i], ot[j], fill_value=0.0))
    for (i, _), mats in Theta.items():
        for th in mats:
            rels.append(Relation(th.copy(), ot[i], ot[i]))
    return FusionGraph(rels), ot


def early_stopping_matches_the_reference_rule(dtypes=('f64',), shards=('runs',)):
    """`stopping`, `stopping_system` and `compute_err` through Dfmf / Dfmc / DfmfTransform on the probe graph (reference
    _dfmf.py:213-221, 301-319; _dfmc.py:271-279, 370-389 with its ((row, col), l) target; fold-in _dfmf.py:367-376, 433-450):
    the fit stops at the iteration at which the oracle, driven with the reference's rule on the same dictionaries and the
    same G0, stops, and returns its factors -- f64 to 1e-9, f32 to its engine tolerance (1e-4), on one device and sharded
    (`shards`: 'rows' / 'owned' need a torch.distributed group; a one-rank group exercises the library's exchanges)."""
    from oracle import dfmf_oracle as orc
    from helpers import relerr
    from skfusion_amd.fusion.decomposition.dfmf import graph_matrices, initial_factors
    checked = 0
    for cls, variant in ((Dfmf, 'dfmf'), (Dfmc, 'dfmc')):
        graph, ot = probe_fusion_graph(masked=(variant == 'dfmc'))
        types = list(graph.object_types)
        rank = {t: int(t.rank) for t in types}
        if variant == 'dfmc':
            R, Theta, M = graph_matrices(graph, with_masks=True)
            assert sum(m is not None for v in M.values() for m in v) == 2
            target = ((ot['t1'], ot['t2']), 0)
        else:
            R, Theta = graph_matrices(graph)
            M = None
            target = (ot['t1'], ot['t3'])
        G0 = initial_factors(R, types, rank, 'random', np.random.RandomState(7), 1)[0]
        # thresholds from the oracle's own error decrements, half-way between those of two consecutive iterations (the
        # decision then has a margin no engine rounding can cross); the reference's errors: target relation and system
        trace = {'t': [], 's': []}

        def watch(G, S, it, R=R, M=M):
            Rw = R
            if variant == 'dfmc':                      # the working copy: masked entries hold the current completion
                Rw = {k: [m.copy() for m in v] for k, v in R.items()}
                for k, masks in M.items():
                    for l, m in enumerate(masks):
                        if m is not None:
                            Rw[k][l][m] = (G[k[0], k[0]] @ S[k][l] @ G[k[1], k[1]].T)[m]
            e = orc.relation_errors(Rw, G, S)
            (i, j), l = target if variant == 'dfmc' else (target, 0)
            trace['t'].append(e[i, j][l])
            trace['s'].append(sum(sum(v) for v in e.values()))
        if variant == 'dfmc':
            orc.dfmc(R, M, Theta, types, rank, max_iter=16, G0=G0, callback=watch)
        else:
            orc.dfmf(R, Theta, types, rank, max_iter=16, G0=G0, callback=watch)
        dt, ds = -np.diff(trace['t']), -np.diff(trace['s'])
        eps_t, eps_s = 0.5 * (dt[9] + dt[10]), 0.5 * (ds[11] + ds[12])
        assert dt[9] > eps_t * 1.005 and dt[10] < eps_t * 0.995 and ds[11] > eps_s * 1.005 and ds[12] < eps_s * 0.995
        for kw, max_iter in ((dict(stopping_system=eps_s), 40), (dict(stopping=(target, eps_t)), 40),
                             (dict(stopping=(target, eps_t), stopping_system=1e-9, compute_err=True), 40)):
            want_seen = []
            okw = dict(kw)
            if variant == 'dfmc':
                Go, So = orc.dfmc(R, M, Theta, types, rank, max_iter=max_iter, G0=G0,
                                  callback=lambda g, s, it: want_seen.append(it), **okw)
            else:
                Go, So = orc.dfmf(R, Theta, types, rank, max_iter=max_iter, G0=G0,
                                  callback=lambda g, s, it: want_seen.append(it), **okw)
            assert 2 < len(want_seen) < max_iter, (variant, kw, len(want_seen))          # the rule fired, and not at once
            for dtype in dtypes:
                for shard in shards:
                    seen = []
                    fuser = cls(max_iter=max_iter, init_type='random', random_state=7, dtype=dtype, shard=shard,
                                callback=lambda g, s, it: seen.append(it), **kw).fuse(graph)
                    assert seen == want_seen, (variant, kw, dtype, shard, len(seen), len(want_seen))
                    tol = 1e-9 if dtype == 'f64' else 1e-4
                    for t in types:
                        assert relerr(fuser.factor(t), Go[t, t]) < tol, (variant, kw, dtype, shard, t.name)
                    checked += 1
        # compute_err alone changes nothing
        a = cls(max_iter=6, init_type='random', random_state=7, compute_err=True).fuse(graph)
        b = cls(max_iter=6, init_type='random', random_state=7).fuse(graph)
        for t in types:
            np.testing.assert_array_equal(a.factor(t), b.factor(t))
    # fold-in: new objects of t1 against the frozen model of a Dfmf fit; stopping_system on the error of the new relations
    graph, ot = probe_fusion_graph(masked=False)
    fuser = Dfmf(max_iter=15, init_type='random', random_state=7).fuse(graph)
    rs = np.random.RandomState(3)
    new = [Relation(rs.rand(9, 30), ot['t1'], ot['t2']), Relation(rs.rand(9, 20) - 0.3, ot['t1'], ot['t3'])]
    from skfusion_amd.fusion.decomposition._init import initialize
    G = {(t, t): fuser.factor(t) for t in graph.object_types}
    S = {(r.row_type, r.col_type): [fuser.backbone(r)] for r in graph.relations if r.row_type != r.col_type}
    Rn = {(r.row_type, r.col_type): [r.data] for r in new}
    rank = {t: int(t.rank) for t in graph.object_types}
    G0 = initialize([ot['t1']], {ot['t1']: 9}, rank, {}, 'random', np.random.RandomState(5))[ot['t1'], ot['t1']]
    # the oracle's fold-in and the reference's rule on the summed error of the new relations (_dfmf.py:433-450); the
    # threshold half-way between two consecutive decrements of that error
    errs = []
    for it in range(16):
        Gh = orc.transform(Rn, {}, ot['t1'], rank, G, S, max_iter=it + 1, G0=G0)
        errs.append(sum(np.linalg.norm(m[0] - Gh @ S[k][0] @ G[k[1], k[1]].T) for k, m in Rn.items()))
    d = -np.diff(errs)
    eps = 0.5 * (d[7] + d[8])
    assert d[7] > eps * 1.005 and d[8] < eps * 0.995
    stop_at = next(it for it in range(2, len(errs)) if errs[it - 2] - errs[it - 1] < eps)
    want = orc.transform(Rn, {}, ot['t1'], rank, G, S, max_iter=stop_at, G0=G0)
    for dtype in dtypes:
        seen = []
        tr = DfmfTransform(max_iter=60, init_type='random', random_state=5, stopping_system=eps, dtype=dtype,
                           callback=lambda g, it: seen.append(it)).transform(ot['t1'], FusionGraph(new), fuser)
        assert 2 < stop_at < 16 and len(seen) == stop_at, (dtype, stop_at, len(seen))
        assert relerr(tr.factor(ot['t1']), want) < (1e-9 if dtype == 'f64' else 1e-4)
    return checked


def fold_in_of_several_runs_shares_launches(dtype='f64', n_new=15, tol=1e-9):
    """DfmfTransform(n_run=3): the fold-ins into the models of the three restarts (reference dfmf.py:191-199: one joblib
    task each) run as plans of ONE set of uploaded relations with every launch shared (skf_iterate_batch) -- the same
    factors, bit for bit, as one fold-in after the other (a callback forces that path), and the oracle's to `tol`."""
    from oracle import dfmf_oracle as orc
    from helpers import relerr
    from skfusion_amd.fusion.decomposition._init import initialize
    rs = np.random.RandomState(21)
    t1, t2, t3 = ObjectType('type1', 7), ObjectType('type2', 5), ObjectType('type3', 6)
    rels = [Relation(rs.rand(40, 30), t1, t2), Relation(rs.rand(40, 25) - 0.2, t1, t3), Relation(rs.rand(30, 25), t2, t3),
            Relation(rs.rand(30, 40), t2, t1)]
    fuser = Dfmf(max_iter=6, init_type='random', random_state=3, n_run=3).fuse(FusionGraph(rels))
    new = lambda: FusionGraph([Relation(np.random.RandomState(5).rand(n_new, 30), t1, t2),          # noqa: E731
                               Relation(np.random.RandomState(6).rand(n_new, 25), t1, t3),
                               Relation(np.random.RandomState(7).rand(30, n_new), t2, t1)])
    shared = DfmfTransform(max_iter=12, init_type='random', random_state=9, n_run=3, dtype=dtype).transform(t1, new(), fuser)
    one_by_one = DfmfTransform(max_iter=12, init_type='random', random_state=9, n_run=3, dtype=dtype,
                               callback=lambda g, it: None).transform(t1, new(), fuser)
    got = list(shared.factor(t1))
    assert len(got) == 3 and got[0].shape == (n_new, 7)
    for a, b in zip(got, one_by_one.factor(t1)):
        np.testing.assert_array_equal(a, b)
    assert not np.allclose(got[0], got[1])
    g = new()
    Rn = {(r.row_type, r.col_type): [r.data] for r in g.relations}
    rank = {t: int(t.rank) for t in (t1, t2, t3)}
    draw = np.random.RandomState(9)
    for run in range(3):
        G = {(t, t): fuser.factor(t, run) for t in (t1, t2, t3)}
        S = {(r.row_type, r.col_type): [fuser.backbone(r, run)] for r in rels}
        G0 = initialize([t1], {t1: n_new}, rank, {}, 'random', draw)[t1, t1]
        want = orc.transform(Rn, {}, t1, rank, G, S, max_iter=12, G0=G0)
        assert relerr(got[run], want) < tol, (dtype, run)


def sharded_fits_of_a_single_process_are_the_plain_fit(dtype='f64', tol=1e-11, max_iter=4):
    """`shard='owned' | 'rows' | 'relations'` on ONE process without a torch.distributed group and without
    SKF_FORCE_COLLECTIVES (the owned mode had fallen to the null communicator of bench.py --emulate-rank,
    which never updates the factors, and returned G0): every mode is then the plain fit -- same factors, same backbones
    (f64: the same launches in another order, 1e-11; f32: its engine tolerance)."""
    import os
    from helpers import relerr
    assert not os.environ.get('SKF_FORCE_COLLECTIVES')
    checked = 0
    for cls, masked in ((Dfmf, False), (Dfmc, True)):
        graph, ot = probe_fusion_graph(masked)
        kw = dict(init_type='random', random_state=np.random.RandomState(5), max_iter=max_iter, dtype=dtype)
        plain = cls(**kw).fuse(graph)
        g0_far = 0.0
        for shard in ('owned', 'rows', 'relations'):
            kw['random_state'] = np.random.RandomState(5)
            fit = cls(shard=shard, **kw).fuse(graph)
            for t in ot.values():
                assert relerr(fit.factor(t), plain.factor(t)) < tol, (cls.__name__, shard, t.name)
            for rel in graph.relations:
                if rel.row_type is not rel.col_type:
                    assert relerr(fit.backbone(rel), plain.backbone(rel)) < max(tol, 1e-9 if dtype == 'f64' else tol)
            checked += 1
        # ... and the plain fit did move away from its initial factors (the comparison above is not G0 against G0)
        first = cls(**dict(kw, max_iter=1, random_state=np.random.RandomState(5))).fuse(graph)
        for t in ot.values():
            g0_far = max(g0_far, relerr(plain.factor(t), first.factor(t)))
        assert g0_far > 1e-3
    return checked


def callback_transport_on_device_views(dtype='f64', tol=1e-11):
    """the callback communicator under a backend that takes device tensors only (nccl) -- its
    all-reduce / all-gather run on the device views of the workspace, not on host copies the backend rejects.  Needs an
    initialised process group; `force_callback` keeps the library from binding RCCL itself."""
    from helpers import golden, probe_graph, relerr, g0_from
    import skfusion_amd._native as nat
    from skfusion_amd.fusion.decomposition import _dfmf
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank = probe_graph(z)
    G0 = g0_from(z, 'dfmf/', types)
    Gp, Sp = _dfmf.dfmf(R, Theta, types, rank, max_iter=3, G0=G0, dtype=dtype)
    n_obj = _dfmf.count_objects(types, R)
    plan = _dfmf.owned_plan(nat.SKF_DFMF, _dfmf.flatten_relations(R, None), _dfmf.flatten_thetas(Theta), types, n_obj, rank,
                            dtype, None, 0, 1)
    try:
        import os
        os.environ['SKF_FORCE_COLLECTIVES'] = '1'
        try:
            assert plan.attach_comm(force_callback=True)
        finally:
            del os.environ['SKF_FORCE_COLLECTIVES']
        for t in types:
            plan.set_factor(t, G0[t, t])
        plan.iterate_dist(3)
        for t in types:
            assert relerr(plan.get_factor(t), Gp[t, t]) < tol
    finally:
        plan.close()


def chained_profiles_match_the_reference_examples(dtype='f64', tol=1e-9, block_rows=17):
    """SURVEY 8 f4 / VERDICT round 4 #4: `chain_profile_blocks` of a fit and of a transformer against
    tests/golden/chain_profiles.npz -- the REFERENCE's fit and fold-in of a four-type graph with one-, two- and three-hop
    paths, the profiles computed by tools/gen_golden.py with the arithmetic of the reference's own examples
    (examples/dicty_chaining.py:40-53: G_row . prod(S) . G_col^T, one type skipped; pharma_chaining.py:43-53: G_row . prod(S)).
    The fitted (G, S) of the golden are attached to this package's classes; paths, backbone products (device, f64), blocks
    (device, `dtype`) and the hstack order are this package's.  Blocks of 17 rows: several blocks, a ragged last one."""
    from helpers import golden, relerr
    z = golden('chain_profiles.npz')
    names = 'ABCD'
    ot = {k: ObjectType(k, int(z['G_' + k].shape[1])) for k in names}
    pairs = [('A', 'B'), ('A', 'C'), ('B', 'C'), ('B', 'D'), ('C', 'D')]
    nan = lambda a, b: np.broadcast_to(np.nan, (z['G_' + a].shape[0], z['G_' + b].shape[0]))     # noqa: E731
    rels = {k: Relation(nan(*k), ot[k[0]], ot[k[1]]) for k in pairs}
    graph = FusionGraph(list(rels.values()) + [Relation(nan('A', 'A'), ot['A'], ot['A'])])
    fuser = Dfmf()
    fuser.fusion_graph = graph
    for k in names:
        fuser.factors_[ot[k]].append(z['G_' + k])
    for k in pairs:
        fuser.backbones_[rels[k]].append(z['S_%s_%s' % k])
    n_new = z['G_new_A'].shape[0]
    transformer = DfmfTransform()
    transformer.target, transformer.fuser = ot['A'], fuser
    transformer.fusion_graph = FusionGraph([Relation(np.broadcast_to(np.nan, (n_new, z['G_' + b].shape[0])), ot['A'], ot[b])
                                            for b in 'BC'])
    transformer.factors_[ot['A']].append(z['G_new_A'])
    order = [ot[k] for k in names]

    said = ['>'.join(t.name for t in path) for _, path in fuser.chain_paths(ot['A'], order)]
    assert said == [str(s) for s in z['paths/all']]
    said = ['>'.join(t.name for t in path) for _, path in fuser.chain_paths(ot['A'], order, skip=[ot['B']])]
    assert said == [str(s) for s in z['paths/skipB']]
    # a three-hop backbone product against the host product of the golden's backbones
    bb = fuser.chain_backbone([ot[k] for k in 'ABCD'])
    assert relerr(bb, z['S_A_B'].dot(z['S_B_C']).dot(z['S_C_D'])) < 1e-13
    worst = 0.0
    for who, tag in ((fuser, 'fit'), (transformer, 'new')):
        for project, skip, key in ((True, [ot['B']], '_project_skipB'), (False, (), '_plain')):
            want = z['profile/' + tag + key]
            seen = 0
            for sl, blk in who.chain_profile_blocks(ot['A'], order, block_rows=block_rows, dtype=dtype, project=project,
                                                    skip=skip):
                assert blk.shape == (sl.stop - sl.start, want.shape[1]) and sl.start == seen
                worst = max(worst, relerr(blk, want[sl]))
                seen = sl.stop
            assert seen == want.shape[0]
            assert relerr(who.chain_profile(ot['A'], order, dtype=dtype, project=project, skip=skip), want) <= max(worst, tol)
    assert worst < tol, worst
    with np.testing.assert_raises(DataFusionError):       # a transformer's profiles start at its target
        list(transformer.chain_profile_blocks(ot['B'], order))
    return worst
