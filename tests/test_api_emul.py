"""Class-layer API on the CPU (host SIMT emulator, few iterations): shapes, generators,
consistency, hooks, error paths.  Full-iteration versions: tests/test_gpu_api.py."""
import pytest

import skfusion_amd._native as nat
from skfusion_amd.fusion import Dfmf, Dfmc
from emul.runtime import emulated_runtime, use_runtime
import api_cases as A


@pytest.fixture(scope='module', autouse=True)
def emul():
    from skfusion_amd._engine import split_clamps
    with use_runtime(emulated_runtime()) as rt:
        yield rt
        assert split_clamps(rt) == 0        # no split-K launch of the module outgrew the scratch its plan sized


@pytest.mark.parametrize('cls', [Dfmf, Dfmc])
def test_shapes_full_rank(cls):
    A.exact_reconstruction(cls, full=False)


def test_non_finite_and_masked_inputs():
    A.non_finite_inputs(full=False)
    A.masked_completion(full=False)


def test_processors_leave_data_untouched():
    A.processors(Dfmf, full=False)


@pytest.mark.parametrize('cls', [Dfmf, Dfmc])
def test_several_runs(cls):
    A.several_runs(cls, full=False)


def test_multiple_relations_per_pair():
    A.multiple_relations(Dfmf, full=False)


def test_pipeline_transform_chain_and_errors():
    A.pipeline_and_transform(full=False)
    A.error_paths()
    A.blockwise_completion()


@pytest.mark.parametrize('cls', [Dfmf, Dfmc])
def test_n_jobs_concurrent_restarts(cls):
    A.n_jobs_concurrent_restarts_equal_sequential(cls)


def test_fill_strategies_on_the_device():
    A.device_fill_strategies()


def test_saved_model_drives_the_engine(tmp_path):
    """f4: fit -> save -> load -> blockwise device completion and fold-in from the loaded model (same case as on the GPU)."""
    A.persistence_round_trip_on_the_engine(tmp_path, 'f64')


def test_early_stopping_through_the_classes():
    """stopping / stopping_system / compute_err through Dfmf, Dfmc and DfmfTransform against the oracle driven with the
    reference's rule (same case as on the GPU, f64)."""
    assert A.early_stopping_matches_the_reference_rule(('f64',), ('runs',)) == 6


def test_early_stopping_inside_sharded_fits_of_a_one_rank_group(monkeypatch):
    """The same decisions with shard='rows' / 'owned' over a one-rank gloo group: the library issues its exchanges through
    the callback communicator, the squared errors are summed over the ranks of the group."""
    import socket
    import torch.distributed as dist
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    try:
        monkeypatch.setenv('SKF_FORCE_COLLECTIVES', '1')
        assert A.early_stopping_matches_the_reference_rule(('f64',), ('rows', 'owned')) == 12
    finally:
        dist.destroy_process_group()


def test_fold_in_of_several_runs_shares_launches():
    A.fold_in_of_several_runs_shares_launches('f64')


def test_chained_profiles_match_the_reference_examples():
    """f4: chained latent profiles (fit and transformer, both example forms) against the golden of the reference run."""
    assert A.chained_profiles_match_the_reference_examples('f64', 1e-12) < 1e-12


def test_sharded_fits_of_a_single_process_are_the_plain_fit(monkeypatch):
    """shard='owned' / 'rows' / 'relations' without a process group and without SKF_FORCE_COLLECTIVES: the plain fit, not G0."""
    monkeypatch.delenv('SKF_FORCE_COLLECTIVES', raising=False)
    assert A.sharded_fits_of_a_single_process_are_the_plain_fit('f64', 1e-11, max_iter=3) == 6


def test_host_rule_for_shared_launches_is_the_librarys_verdict_on_boundary_graphs():
    """`shared_launches` decides on the host whether restarts will share launches; its limits now come from the
    library (skf_small_graph_limits).  On graphs that sit on either side of every limit -- rank 64 / 65, 8192 / 8193 objects,
    a constraint with n^2/16 and n^2/16 + 1 non-zeros, an all-zero constraint -- the host verdict equals skf_plan_batchable
    of the bound plan."""
    import numpy as np
    from skfusion_amd.fusion import Relation, ObjectType, FusionGraph
    from skfusion_amd.fusion.decomposition.dfmf import shared_launches, graph_matrices
    from skfusion_amd._engine import DevicePlan, flatten_relations, flatten_thetas, count_objects

    def verdicts(rank_a, n_b, theta_nnz):
        rs = np.random.RandomState(0)
        a, b = ObjectType('a', rank_a), ObjectType('b', 3)
        n_a = 32
        rels = [Relation(rs.rand(n_a, n_b), a, b)]
        if theta_nnz is not None:
            th = np.zeros((n_a, n_a))
            th.flat[:theta_nnz] = -0.1
            rels.append(Relation(th, a, a))
        graph = FusionGraph(rels)
        fuser = Dfmf(n_run=3, max_iter=1, init_type='random', random_state=0)
        fuser.fusion_graph = graph
        R, Theta = graph_matrices(graph)
        types = list(graph.object_types)
        plan = DevicePlan(types, count_objects(types, R), {t: int(t.rank) for t in types}, flatten_relations(R, None),
                          flatten_thetas(Theta), nat.SKF_DFMF, dtype='f64')
        try:
            return shared_launches(fuser), plan.batchable()
        finally:
            plan.close()
    cases = [(64, 40, None), (65, 40, None), (8, 8192, None), (8, 8193, None), (8, 40, 64), (8, 40, 65), (8, 40, 0), (8, 40, 1)]
    seen = set()
    for case in cases:
        host, lib = verdicts(*case)
        assert host == lib, (case, host, lib)
        seen.add(host)
    assert seen == {True, False}
