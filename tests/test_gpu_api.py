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
