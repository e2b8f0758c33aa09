"""Class-layer API on the MI355X with the reference's own iteration counts and assertions
(exact reconstruction at full rank to 7 decimals, fold-in MSE < 1e-5, ...)."""
import pytest

from skfusion_amd.fusion import Dfmf, Dfmc
import api_cases as A

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cls', [Dfmf, Dfmc])
def test_exact_reconstruction_at_full_rank(cls):
    A.exact_reconstruction(cls, full=True)


def test_non_finite_inputs():
    A.non_finite_inputs(full=True)


def test_masked_completion():
    A.masked_completion(full=True)


@pytest.mark.parametrize('cls', [Dfmf, Dfmc])
def test_processors(cls):
    A.processors(cls, full=True)


@pytest.mark.parametrize('cls', [Dfmf, Dfmc])
def test_several_runs_rank_deficient(cls):
    A.several_runs(cls, full=True)


@pytest.mark.parametrize('cls', [Dfmf, Dfmc])
def test_multiple_relations(cls):
    A.multiple_relations(cls, full=True)


def test_pipeline_and_fold_in():
    A.pipeline_and_transform(full=True)
    A.fold_in_recovers_known_rows(full=True)
    A.error_paths()
    A.blockwise_completion()


def test_early_stopping_through_the_classes_single_device_and_sharded(monkeypatch):
    """`stopping`, `stopping_system`, `compute_err` through Dfmf / Dfmc / DfmfTransform on the probe graph
    ON THE HARDWARE: the same stopping iteration and the same final factors as the oracle driven with the reference's rule
    (_dfmf.py:213-221, 301-319; _dfmc.py:370-389), f64 and f32, on one device and with shard='rows' / 'owned' over a one-rank
    RCCL group (the library's own exchanges, squared errors summed over the ranks)."""
    import socket
    import torch
    import torch.distributed as dist
    assert A.early_stopping_matches_the_reference_rule(('f64', 'f32'), ('runs',)) == 12
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    try:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                                device_id=torch.device('cuda', torch.cuda.current_device()))
    except Exception as exc:                                  # pragma: no cover
        pytest.skip('no one-rank RCCL group on this box: %r' % (exc,))
    try:
        monkeypatch.setenv('SKF_FORCE_COLLECTIVES', '1')
        assert A.early_stopping_matches_the_reference_rule(('f64', 'f32'), ('rows', 'owned')) == 24
    finally:
        dist.destroy_process_group()


# measured (MI355X): f64 1.4e-16 (rounding level: the bound is a floor of ~10 ulp), f32 / bf16 (f32 masters) 6.0e-8
@pytest.mark.parametrize('dtype,tol', [('f64', 2e-15), ('f32', 6e-7), ('bf16', 6e-7)])
def test_chained_profiles_match_the_reference_examples(dtype, tol):
    """chained latent profiles on the device -- backbone products in f64, blocks through
    the matrix-core GEMM of the engine's master type -- against the golden of the reference's fit / fold-in with the
    arithmetic of its own examples (dicty_chaining.py:40-53, pharma_chaining.py:43-53)."""
    from helpers import within
    within(A.chained_profiles_match_the_reference_examples(dtype, tol), tol, 'chained profiles vs reference golden, ' + dtype)


@pytest.mark.parametrize('dtype,tol', [('f64', 1e-11), ('f32', 1e-5)])
def test_sharded_fits_of_a_single_process_are_the_plain_fit(dtype, tol, monkeypatch):
    """shard='owned' on one process without a group returned the unfitted G0 (null communicator).  Every
    sharded mode of a single process is the plain fit; also with the nccl fall-back transport (collectives of the callback
    communicator on device views) over a one-rank RCCL group."""
    import socket
    import torch
    import torch.distributed as dist
    monkeypatch.delenv('SKF_FORCE_COLLECTIVES', raising=False)
    assert A.sharded_fits_of_a_single_process_are_the_plain_fit(dtype, tol) == 6
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    try:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                                device_id=torch.device('cuda', torch.cuda.current_device()))
    except Exception as exc:                                  # pragma: no cover
        pytest.skip('no one-rank RCCL group on this box: %r' % (exc,))
    try:
        A.callback_transport_on_device_views(dtype, tol)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('dtype,tol', [('f64', 1e-9), ('f32', 1e-5), ('bf16', 1e-2)])
def test_fold_in_of_several_runs_shares_launches(dtype, tol):
    A.fold_in_of_several_runs_shares_launches(dtype, tol=tol)


def test_f32_engine_reaches_the_same_fixed_point():
    import numpy as np
    from skfusion_amd.fusion import Relation, ObjectType, FusionGraph
    rnds = np.random.RandomState(0)
    R12 = rnds.rand(50, 30)
    t1, t2 = ObjectType('type1', 50), ObjectType('type2', 30)
    rel = Relation(R12, t1, t2)
    fuser = Dfmf(init_type='random', random_state=rnds, dtype='f32').fuse(FusionGraph([rel]))
    np.testing.assert_almost_equal(fuser.complete(rel), R12, decimal=4)


@pytest.mark.parametrize('cls', [Dfmf, Dfmc])
def test_n_jobs_concurrent_restarts(cls):
    A.n_jobs_concurrent_restarts_equal_sequential(cls)


def test_fill_strategies_on_the_device():
    A.device_fill_strategies()


@pytest.mark.parametrize('dtype', ['f32', 'f64'])
def test_saved_model_drives_the_device_path(tmp_path, dtype):
    """f4 on the device: fit on the HIP engine -> save -> load -> complete_blocks(device=True) and DfmfTransform from the
    loaded model, bit for bit."""
    A.persistence_round_trip_on_the_engine(tmp_path, dtype)


def test_integration_stub_binds_the_library_and_reproduces_the_reference_golden():
    """INTEGRATION.md's reference-side stub (what a maintainer would put behind dfmf.py:14-15) is extracted from the
    document, bound to the built library with nothing but ctypes + torch, and must reproduce the reference's own README
    run (tests/golden/c1_readme_dfmf.npz: random_vcol, RandomState(0), 100 iterations) to 1e-9."""
    import os
    import re
    import sys
    import types
    import numpy as np
    from helpers import golden, readme_graph, relerr, within
    from oracle import dfmf_oracle as orc
    import skfusion_amd._native as nat
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    code = re.search(r"```python\n(# skfusion/fusion/decomposition/_hip.py.*?)```", text, re.S).group(1)
    assert 'libskfusion_hip.so' in code and 'skf_plan_create' in code
    code = code.replace('"libskfusion_hip.so"', repr(nat.LIB_PATH))
    # the stub lives in the reference's package: `from ._dfmf import count_objects, initialize` are the REFERENCE's helpers
    # (_dfmf.py:95-124, _init.py:6-61); here their pinned restatements stand in for them
    pkg = types.ModuleType('refside')
    pkg.__path__ = []
    helpers_mod = types.ModuleType('refside._dfmf')
    helpers_mod.count_objects, helpers_mod.initialize = orc.count_objects, orc.initialize
    sys.modules['refside'], sys.modules['refside._dfmf'] = pkg, helpers_mod
    try:
        mod = types.ModuleType('refside._hip')
        mod.__package__ = 'refside'
        exec(compile(code, 'INTEGRATION.md:_hip.py', 'exec'), mod.__dict__)
        z = golden('c1_readme_dfmf.npz')
        R, typs, rank = readme_graph()
        G, S = mod.dfmf(R=R, Theta={}, obj_types=typs, obj_type2rank=rank, max_iter=100, init_type='random_vcol',
                        random_state=np.random.RandomState(0))
    finally:
        sys.modules.pop('refside', None)
        sys.modules.pop('refside._dfmf', None)
    for t in typs:
        within(relerr(G[t, t], z['random_vcol/G_%s_it99' % t]), 1e-11, 'INTEGRATION.md stub: G_%s after 100 iterations vs reference golden' % t)
    for (i, j) in R:
        within(relerr(S[i, j][0], z['random_vcol/S_%s_%s_0_it99' % (i, j)]), 3e-10, 'INTEGRATION.md stub: S_%s_%s vs reference golden' % (i, j))
