"""N > 1 path on CPU: two processes (gloo), restarts sharded round robin, results gathered;
every rank must end up with the same factors as a single-process fit.  The arithmetic runs in
the host SIMT emulator (no GPU in the build container)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from helpers import C5_RELATIONS as C5_REL, C5_TYPES      # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _graph():
    from skfusion_amd.fusion import Relation, ObjectType, FusionGraph
    rs = np.random.RandomState(3)
    t1, t2, t3 = ObjectType('a', 4), ObjectType('b', 3), ObjectType('c', 2)
    rels = [Relation(rs.rand(12, 9), t1, t2), Relation(rs.rand(12, 7), t1, t3)]
    return FusionGraph(rels), (t1, t2, t3), rels


def _fit():
    from skfusion_amd.fusion import Dfmf
    g, types, rels = _graph()
    fuser = Dfmf(max_iter=3, init_type='random_vcol', n_run=3, random_state=7).fuse(g)
    return [np.concatenate([f.ravel() for f in fuser.factor(t)]) for t in types] + \
           [np.concatenate([s.ravel() for s in fuser.backbone(r)]) for r in rels]


def _probe_sharded(shard):
    """Multi-relation / Theta / mask graph, relations partitioned over the ranks."""
    from helpers import golden, probe_graph, g0_from
    from skfusion_amd.fusion.decomposition import _dfmf, _dfmc
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank = probe_graph(z)
    G, S = _dfmf.dfmf(R, Theta, types, rank, max_iter=10, G0=g0_from(z, 'dfmf/', types), shard=shard)
    Gc, Sc = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=10, G0=g0_from(z, 'dfmc/', types), shard=shard)
    out = [G[t, t] for t in types] + [s for k in sorted(S) for s in S[k]]
    out += [Gc[t, t] for t in types] + [s for k in sorted(Sc) for s in Sc[k]]
    return out


def _c5_sharded(shard='relations'):
    """BASELINE config 5 (scaled): MovieLens-style Dfmc, its 6 relations / 3 constraints over the ranks."""
    from helpers import golden, movielens_style_graph, g0_from
    from skfusion_amd.fusion.decomposition import _dfmc
    z = golden('c5_movielens_scaled.npz')
    R, M, Theta, types, rank = movielens_style_graph()
    G, S = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=2, G0=g0_from(z, 'dfmc/', types), shard=shard)
    return [G[t, t] for t in types] + [S[i, j][0] for i, j, _, _ in C5_REL]


def _probe_stopping(shard):
    """callback / compute_err / stopping_system inside a sharded fit: every rank sees the errors of ALL relations
    (summed over the ranks) and therefore stops at the same iteration as the single-device fit."""
    from helpers import golden, probe_graph, g0_from
    from skfusion_amd.fusion.decomposition import _dfmf
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank = probe_graph(z)
    seen = []
    G, S = _dfmf.dfmf(R, Theta, types, rank, max_iter=40, G0=g0_from(z, 'dfmf/', types), shard=shard,
                      stopping_system=2e-2, callback=lambda g, s_, it: seen.append(it))
    seen2 = []
    G2, S2 = _dfmf.dfmf(R, Theta, types, rank, max_iter=12, G0=g0_from(z, 'dfmf/', types), shard=shard,
                        stopping=(('t1', 't3'), 1e-3), callback=lambda g, s_, it: seen2.append(it))
    return [np.array(seen, dtype=np.int64), np.array(seen2, dtype=np.int64)] + [G[t, t] for t in types] + \
           [G2[t, t] for t in types]


def _library_exchange_against_python_exchange(shard):
    """The same sharded plan iterated twice: exchanges issued by the library (skf_iterate_dist through a callback
    communicator: all-reduce of W / Q, reduce-scatter of E and D, update of the owned range, all-gather of G) and exchanges
    issued from Python between the stages (all-reduce of everything).  Both must give the same factors."""
    import skfusion_amd._native as nat
    from helpers import golden, probe_graph, g0_from
    from skfusion_amd._engine import DevicePlan, flatten_relations, flatten_thetas, count_objects
    from skfusion_amd._distributed import partition_relations, world
    from skfusion_amd.fusion.decomposition._dfmf import row_block_plan
    z = golden('probe_multirel.npz')
    R, Theta, M, types, rank_of = probe_graph(z)
    n = count_objects(types, R)
    rel, th = flatten_relations(R, M), flatten_thetas(Theta)
    rank, size = world()
    out = []
    for use_library in (True, False):
        if shard == 'rows':
            plan = row_block_plan(nat.SKF_DFMC, rel, th, types, n, rank_of, 'f64', None, rank, size)
        else:
            owner, th_owner = partition_relations(rel, th, n, rank_of)
            plan = DevicePlan(types, n, rank_of, [r for r, o in zip(rel, owner) if o == rank],
                              [t for t, o in zip(th, th_owner) if o == rank], nat.SKF_DFMC)
        for t in types:
            plan.set_factor(t, z['dfmc/G0_%s' % t])
        if use_library:
            assert plan.attach_comm() and plan.exchange_bytes(size) > 0 and plan.exchange_bytes(1) == 0
            plan.iterate_dist(6)
        elif shard == 'rows':
            plan.iterate_rows(6)
        else:
            plan.iterate_sharded(6)
        out.append([plan.get_factor(t) for t in types])
        plan.close()
    return [a for a in out[0]] + [b for b in out[1]]


def _bf16_owned():
    """The bf16 engine under row ownership over a real process group: the bf16 operand rows of the types without a constraint
    are gathered as BYTES through the callback communicator (gloo has no 16-bit integer type), the constrained type's f32
    rows as floats."""
    from test_owned_sharding import _wide_graph
    from skfusion_amd.fusion.decomposition import _dfmc
    R, M, Theta, types, rank, G0 = _wide_graph()
    G, S = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=2, G0=G0, dtype='bf16', shard='owned')
    return [G[t, t] for t in types] + [S[k][0] for k in sorted(S)]


def _rccl_rendezvous(rank, out):
    """The failure paths of the RCCL rendezvous (_engine._shared_rccl_comm) over a real process group, with the library's two
    entry points replaced by stand-ins: (a) ONE rank cannot bind RCCL (skf_comm_unique_id fails there) -- every rank must
    come back with None, nobody enters ncclCommInitRank; (b) every rank has the id but skf_comm_create fails on one -- again
    None everywhere, and the decision is cached per group; (c) both succeed -- the same handle is handed out twice."""
    import ctypes as C
    import torch.distributed as dist
    import skfusion_amd._native as nat
    import skfusion_amd._engine as eng

    class Stub(object):
        def __init__(self, fail_id=False, fail_create=False):
            self.fail_id, self.fail_create, self.calls = fail_id, fail_create, []
            self.lib = self
            self.mem = None

        def skf_comm_destroy(self, comm):
            self.calls.append('destroy')
            return 0

        def call(self, fname, *args):
            self.calls.append(fname)
            if fname == 'skf_comm_unique_id':
                if self.fail_id:
                    raise nat.SkfNativeError(-5, 'RCCL not found (stand-in)')
                C.memmove(args[0], b'x' * 128, 128)
            elif fname == 'skf_comm_create':
                if self.fail_create:
                    raise nat.SkfNativeError(-5, 'ncclCommInitRank failed (stand-in)')
                C.cast(args[3], C.POINTER(C.c_void_p))[0] = 0x1234
            else:
                raise AssertionError(fname)
    seen = []
    for case, stub in (('a', Stub(fail_id=(rank == 1))), ('b', Stub(fail_create=(rank == 0))), ('c', Stub())):
        eng._RCCL.clear()
        got = eng._shared_rccl_comm(stub, dist)
        again = eng._shared_rccl_comm(stub, dist)              # cached for this group: no second rendezvous
        n_create = stub.calls.count('skf_comm_create')
        if case == 'a':
            assert got is None and again is None and n_create == 0, (case, rank, stub.calls)
        elif case == 'b':
            assert got is None and again is None and n_create == 1, (case, rank, stub.calls)
        else:
            assert got is not None and again is got and n_create == 1, (case, rank, stub.calls)
        seen.append(n_create)
    eng._RCCL.clear()
    np.savez(os.path.join(out, 'rccl%d.npz' % rank), np.array(seen))


def _worker(rank, world, port, out, what):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import skfusion_amd._native as nat
    from skfusion_amd._distributed import my_runs, world as dist_world
    from emul.runtime import emulated_runtime, use_runtime
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        assert dist_world() == (rank, world)
        assert my_runs(3) == [k for k in range(3) if k % world == rank]
        with use_runtime(emulated_runtime()):
            if what == 'rccl':
                _rccl_rendezvous(rank, out)
            elif what == 'runs':
                np.savez(os.path.join(out, 'rank%d.npz' % rank), *_fit())
            elif what.startswith('stop:'):
                np.savez(os.path.join(out, 'stop%d.npz' % rank), *_probe_stopping(what[5:]))
            elif what == 'bf16:owned':
                np.savez(os.path.join(out, 'bf16_%d.npz' % rank), *_bf16_owned())
            elif what.startswith('lib:'):
                np.savez(os.path.join(out, 'lib%d.npz' % rank), *_library_exchange_against_python_exchange(what[4:]))
            else:
                np.savez(os.path.join(out, 'shard%d.npz' % rank), *_probe_sharded(what))
                np.savez(os.path.join(out, 'c5_%d.npz' % rank), *_c5_sharded(what))
    finally:
        dist.destroy_process_group()


def test_restarts_sharded_over_two_gloo_ranks(tmp_path):
    import torch.multiprocessing as mp
    import skfusion_amd._native as nat
    from emul.runtime import emulated_runtime, use_runtime, build
    build()                                   # compile once, before the workers race for it
    with use_runtime(emulated_runtime()):
        single = _fit()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), 'runs'), nprocs=2, join=True)
    for rank in range(2):
        z = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % rank))
        for k, want in enumerate(single):
            np.testing.assert_array_equal(z['arr_%d' % k], want)


@pytest.mark.parametrize('shard,world', [('relations', 2), ('rows', 2), ('owned', 2), ('owned', 3)])
def test_one_fit_sharded_over_gloo_ranks_matches_golden(tmp_path, shard, world):
    """One fit over 2 (3) ranks -- whole relations + constraints partitioned (E/D all-reduced every
    iteration), balanced row blocks of the relations (W, Q, E/D all-reduced between the stages), or the rows of every type
    with the matching rows of its relations (`owned`: reduce-scatter of the partial Q, all-gather of the updated rows,
    issued by the library through the callback communicator):
    every rank must reproduce the reference goldens (iteration 10 of the probe graph, DFMF and
    DFMC; iteration 2 of the MovieLens-style config 5) to 1e-9 -- the bar of the single-device engine."""
    import torch.multiprocessing as mp
    from emul.runtime import build
    from helpers import golden, TYPES, relerr
    build()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), shard), nprocs=world, join=True)
    z = golden('probe_multirel.npz')
    pairs = [('t1', 't2', 0), ('t1', 't2', 1), ('t1', 't3', 0), ('t2', 't3', 0)]
    for rank in range(world):
        a = np.load(os.path.join(str(tmp_path), 'shard%d.npz' % rank))
        arrs = [a['arr_%d' % k] for k in range(len(a.files))]
        for v, variant in ((0, 'dfmf'), (7, 'dfmc')):
            for k, t in enumerate(TYPES):
                assert relerr(arrs[v + k], z['%s/G_%s_it9' % (variant, t)]) < 1e-9
            for k, (i, j, l) in enumerate(pairs):
                assert relerr(arrs[v + 3 + k], z['%s/S_%s_%s_%d_it9' % (variant, i, j, l)]) < 1e-9
    # config 5 (MovieLens-style Dfmc, 6 relations + 3 constraints over 2 ranks): iteration 2 of the golden
    z5 = golden('c5_movielens_scaled.npz')
    for rank in range(world):
        a = np.load(os.path.join(str(tmp_path), 'c5_%d.npz' % rank))
        arrs = [a['arr_%d' % k] for k in range(len(a.files))]
        for k, t in enumerate(C5_TYPES):
            assert relerr(arrs[k], z5['dfmc/G_%s_it1' % t]) < 1e-9
        for k, (i, j, _, _) in enumerate(C5_REL):
            assert relerr(arrs[len(C5_TYPES) + k], z5['dfmc/S_%s_%s_0_it1' % (i, j)]) < 1e-9


@pytest.mark.parametrize('shard', ['relations', 'rows', 'owned'])
def test_stopping_and_callback_inside_a_sharded_fit(tmp_path, shard):
    """`stopping`, `stopping_system`, `compute_err` and `callback` under shard='relations' / 'rows' (the reference
    supports them on every path, _dfmf.py:213-221, 301-322): both ranks stop at the iteration the single-device fit
    stops at and return its factors."""
    import torch.multiprocessing as mp
    import skfusion_amd._native as nat
    from emul.runtime import emulated_runtime, use_runtime, build
    from helpers import relerr
    build()
    with use_runtime(emulated_runtime()):
        single = _probe_stopping(None)
    assert 2 < len(single[0]) < 40 and 2 < len(single[1]) <= 12        # stopping_system fired early
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), 'stop:' + shard), nprocs=2, join=True)
    for rank in range(2):
        a = np.load(os.path.join(str(tmp_path), 'stop%d.npz' % rank))
        np.testing.assert_array_equal(a['arr_0'], single[0])
        np.testing.assert_array_equal(a['arr_1'], single[1])
        for k in range(2, len(single)):
            assert relerr(a['arr_%d' % k], single[k]) < 1e-9


@pytest.mark.parametrize('shard', ['relations', 'rows'])
def test_exchanges_issued_by_the_library_match_the_python_exchanges(tmp_path, shard):
    """skf_comm_create_callback + skf_plan_set_comm + skf_iterate_dist over two gloo ranks: reduce-scatter(E), reduce-scatter(D),
    update of the owned element range, all-gather(G) -- against the all-reduce of both accumulators issued from Python."""
    import torch.multiprocessing as mp
    from emul.runtime import build
    from helpers import golden, TYPES, relerr
    build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), 'lib:' + shard), nprocs=2, join=True)
    z = golden('probe_multirel.npz')
    for rank in range(2):
        a = np.load(os.path.join(str(tmp_path), 'lib%d.npz' % rank))
        for k, t in enumerate(TYPES):
            assert relerr(a['arr_%d' % k], a['arr_%d' % (k + 3)]) < 1e-12
            assert relerr(a['arr_%d' % k], z['dfmc/G_%s_it5' % t]) < 1e-9 if 'dfmc/G_%s_it5' % t in z.files else True


def test_bf16_engine_sharded_by_ownership_over_two_gloo_ranks(tmp_path):
    """bf16 rows through the callback communicator of a gloo group (bytes): both ranks end with the same factors, those of
    the single-process bf16 fit up to the order of the partial sums."""
    import torch.multiprocessing as mp
    from emul.runtime import emulated_runtime, use_runtime, build
    from helpers import relerr
    from test_owned_sharding import _wide_graph
    from skfusion_amd.fusion.decomposition import _dfmc
    build()
    with use_runtime(emulated_runtime()):
        R, M, Theta, types, rank, G0 = _wide_graph()
        Gs, Ss = _dfmc.dfmc(R, M, Theta, types, rank, max_iter=2, G0=G0, dtype='bf16')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), 'bf16:owned'), nprocs=2, join=True)
    a = [np.load(os.path.join(str(tmp_path), 'bf16_%d.npz' % r)) for r in range(2)]
    for k, t in enumerate(types):
        np.testing.assert_array_equal(a[0]['arr_%d' % k], a[1]['arr_%d' % k])
        assert relerr(a[0]['arr_%d' % k], Gs[t, t]) < 5e-3


def test_rccl_rendezvous_failure_paths_agree_on_every_rank(tmp_path):
    """Round 6: the first multi-rank RCCL run must not hang on a rank that cannot bind the library or
    whose ncclCommInitRank fails -- all ranks agree on the callback transport instead.  Two gloo ranks, the two library entry
    points replaced by stand-ins; and the bound on a rendezvous that never returns (one rank, SKF_COMM_TIMEOUT)."""
    import torch.multiprocessing as mp
    from emul.runtime import build
    build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), 'rccl'), nprocs=2, join=True)
    for rank in range(2):
        z = np.load(os.path.join(str(tmp_path), 'rccl%d.npz' % rank))
        np.testing.assert_array_equal(z['arr_0'], [0, 1, 1])


def test_rccl_rendezvous_is_bounded_in_time(monkeypatch):
    """A rank still inside ncclCommInitRank after SKF_COMM_TIMEOUT seconds raises (the caller treats it as fatal for the
    process group) instead of waiting for the transport's own watchdog."""
    import time
    import ctypes as C
    import skfusion_amd._engine as eng

    class FakeDist(object):
        @staticmethod
        def get_rank():
            return 0

        @staticmethod
        def get_world_size():
            return 1

        @staticmethod
        def get_backend():
            return 'gloo'

        @staticmethod
        def all_reduce(t, op=None):
            return None

        class ReduceOp(object):
            MIN = 'min'

        @staticmethod
        def broadcast_object_list(objs, src=0):
            return None

    class Slow(object):
        lib = None

        def call(self, fname, *args):
            if fname == 'skf_comm_unique_id':
                C.memmove(args[0], b'y' * 128, 128)
            else:
                time.sleep(3.0)
    monkeypatch.setenv('SKF_COMM_TIMEOUT', '0.3')
    monkeypatch.setattr(eng, '_group_token', lambda dist: 'fake-group')
    eng._RCCL.clear()
    t0 = time.time()
    with pytest.raises(RuntimeError, match='still waits in ncclCommInitRank'):
        eng._shared_rccl_comm(Slow(), FakeDist)
    assert time.time() - t0 < 2.0
    eng._RCCL.clear()
