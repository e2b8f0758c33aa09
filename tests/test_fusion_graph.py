"""Host-side data model: graph bookkeeping (the properties reference
skfusion/tests/test_fusion_graph.py:29-137 pins), the fill strategies against outputs captured
from the reference (tests/golden/fill_strategies.npz) and the graph -> (R, Theta, M) seam glue."""
import warnings

import numpy as np
import pytest

from skfusion_amd.fusion import FusionGraph, Relation, ObjectType, DataFusionError
from skfusion_amd.fusion.decomposition.dfmf import graph_matrices
from helpers import golden


@pytest.fixture
def world():
    X = np.random.RandomState(0).rand(30, 30)
    t = [ObjectType('Type %d' % k, 10) for k in range(1, 6)]
    t1, t2, t3, t4, t5 = t
    rels = [Relation(X, t1, t2, name='Test2'), Relation(X, t1, t2), Relation(X, t2, t3),
            Relation(X, t3, t4), Relation(X, t4, t5), Relation(X, t3, t5), Relation(X, t5, t1),
            Relation(X, t4, t4), Relation(X, t4, t4, name='Test3'), Relation(X, t5, t5)]
    return t, rels


def test_lookup_counts_and_removal(world):
    t, rels = world
    g = FusionGraph()
    g.add_relations_from(rels)
    assert g['Test2'] == rels[0] and g['Test3'] == rels[8]
    assert g.get_relation('Test3') is rels[8]
    assert (g.n_object_types, g.n_relations) == (5, 10)
    g.remove_relation(rels[6])
    assert (g.n_object_types, g.n_relations) == (5, 9)
    g.remove_relations_from([rels[9], rels[4], rels[5]])      # Type 5 loses its last relation
    assert (g.n_object_types, g.n_relations) == (4, 6)
    with pytest.raises(DataFusionError):
        g.get_relation('nope')
    with pytest.raises(DataFusionError):
        g.get_object_type('Type 5')


def test_neighbourhood_queries(world):
    t, rels = world
    t1, t2, t3, t4, t5 = t
    g = FusionGraph(rels)
    assert set(g.in_relations(t1)) == {rels[6]}
    assert set(g.out_relations(t1)) == set(rels[:2])
    assert set(g.out_relations(t4)) == {rels[4], rels[7], rels[8]}
    assert g.get_object_type('Type 1') == t1
    assert list(g.get_relations(t1, t2)) == rels[:2]
    assert g[t1][t2] == rels[:2]
    assert len(list(g.out_relations(t4))) == sum(len(v) for v in g[t4].values())
    assert set(g.out_neighbors(t3)) == {t4, t5} and set(g.in_neighbors(t1)) == {t5}
    with pytest.raises(DataFusionError):
        list(g.out_relations(ObjectType('stranger')))
    with pytest.raises(DataFusionError):
        g.get_relations(t1, ObjectType('stranger'))


def test_single_relation_and_self_loop_removal(world):
    t, rels = world
    g = FusionGraph()
    g.add_relation(rels[0])
    assert (g.n_relations, g.n_object_types) == (1, 2)
    g.remove_relation(rels[0])
    assert (g.n_relations, g.n_object_types) == (0, 0)
    g.add_relation(rels[-1])                                   # Type 5 -> Type 5
    assert (g.n_relations, g.n_object_types) == (1, 1)
    g.remove_relation(rels[-1])
    assert (g.n_relations, g.n_object_types) == (0, 0)


def test_names_and_metadata(world):
    t, _ = world
    t1, t2, t3 = t[:3]
    X = np.random.RandomState(0).rand(10, 10)
    a, b = list('ABCDEFGHIJ'), list('KLMNOPQRST')
    m1 = [{'a': x} for x in a]
    m2 = [{'b': x} for x in '0123456789']
    m2b = [{'d': x} for x in '0123456789']
    g = FusionGraph([
        Relation(X, name='r', row_type=t1, row_names=a, col_type=t2, col_names=b, row_metadata=m1,
                 col_metadata=m2),
        Relation(X, name='r2', row_type=t2, row_names=b, col_type=t3, row_metadata=m2b)])
    assert g.get_names(t1) == a and g.get_names('Type 2') == b
    assert g.get_names(t3) == [str(k) for k in range(10)]
    assert g.get_metadata(t1) == m1
    assert g.get_metadata(t2) == [dict(x, **y) for x, y in zip(m2, m2b)]
    assert g.get_metadata(t3) == [{}] * 10


def test_object_type_and_relation_identity():
    a, b = ObjectType('x', 3), ObjectType('x', 7)
    assert a == b and hash(a) == hash(b) and str(a) == 'x' and repr(a) == 'ObjectType("x")'
    X = np.zeros((2, 2))
    r1, r2 = Relation(X, a, ObjectType('y')), Relation(X, a, ObjectType('y'))
    assert r1 != r2 and r1 == r1 and a in r1 and ObjectType('z') not in r1
    named = Relation(X, a, ObjectType('y'), name='n', custom=5)
    assert named.custom == 5 and '"n"' in repr(named)
    with pytest.raises(NotImplementedError):
        FusionGraph([r1]).draw_graphviz('x.pdf')


@pytest.mark.parametrize('tag', ['masked', 'plain', 'finite', 'corner', 'plaincorner', 'nomask', 'nomaskfinite', 'maskedinf'])
@pytest.mark.parametrize('fv', ['mean', 'row_mean', 'col_mean', 0.5])
def test_fill_strategies_match_reference_outputs(tag, fv):
    z = golden('fill_strategies.npz')
    if tag.startswith('plain'):
        arr = z[tag].copy()
    else:
        arr = np.ma.MaskedArray(z[tag + '_data'].copy(), mask=z[tag + '_mask'].copy())
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        f = Relation(arr, ObjectType('a'), ObjectType('b'), fill_value=fv).filled()
    key = '%s/%s' % (tag, fv)
    np.testing.assert_array_equal(np.ma.getdata(f), z[key + '/data'])
    np.testing.assert_array_equal(np.ma.getmaskarray(f), z[key + '/mask'])
    assert bool(np.ma.is_masked(f)) == bool(z[key + '/is_masked'])


def test_graph_to_seam_matrices():
    """dfmf.py:70-85 / dfmc.py:70-93: R/Theta split, walking order, fill -> preprocess, masks."""
    rs = np.random.RandomState(1)
    a, b, c = ObjectType('a', 2), ObjectType('b', 3), ObjectType('c', 4)
    Rab = rs.rand(4, 5)
    Rab2 = np.ma.masked_greater(rs.rand(4, 5), 0.5)
    Rbc = rs.rand(5, 6)
    Taa = rs.rand(4, 4)
    rels = [Relation(Rbc, b, c), Relation(Rab, a, b, preprocessor=lambda x: 2 * x),
            Relation(Rab2, a, b, fill_value=0.25), Relation(Taa, a, a)]
    g = FusionGraph(rels)
    R, Theta, M = graph_matrices(g, with_masks=True)
    assert list(R.keys()) == [(b, c), (a, b)]          # product(object_types) in insertion order
    np.testing.assert_array_equal(R[a, b][0], 2 * Rab)
    assert M[a, b][0] is None and M[b, c] == [None]
    np.testing.assert_array_equal(M[a, b][1], Rab2.mask)
    assert not isinstance(R[a, b][1], np.ma.MaskedArray)
    np.testing.assert_array_equal(R[a, b][1][Rab2.mask], 0.25)
    assert list(Theta.keys()) == [(a, a)] and Theta[a, a][0] is not None
    R2, T2 = graph_matrices(g)
    assert set(R2) == set(R) and set(T2) == set(Theta)


def test_restarts_share_the_column_rankings_of_random_c_without_changing_the_stream():
    """`initial_factors` keeps the column rankings of `random_c` (they draw nothing from the random stream) across the restarts
    of one fit: the factors of every restart are those of n_run separate `initialize` calls on one RandomState (reference
    dfmf.py:87-95 with n_jobs = 1: one stream, consumed restart after restart), also for a square relation and its transpose."""
    from skfusion_amd.fusion.decomposition.dfmf import initial_factors
    from skfusion_amd.fusion.decomposition._init import initialize
    rs = np.random.RandomState(3)
    a, b, c = ObjectType('a', 3), ObjectType('b', 4), ObjectType('c', 2)
    R = {(a, b): [rs.rand(12, 12)], (b, a): [rs.rand(12, 12)], (a, c): [rs.rand(12, 9)]}
    rank = {a: 3, b: 4, c: 2}
    got = initial_factors(R, [a, b, c], rank, 'random_c', np.random.RandomState(11), 4)
    state = np.random.RandomState(11)
    first = {k: np.asarray(v[0], dtype=float) for k, v in R.items()}
    n_obj = {a: 12, b: 12, c: 9}
    for run in range(4):
        want = initialize([a, b, c], n_obj, rank, first, 'random_c', state)
        for t in (a, b, c):
            np.testing.assert_array_equal(got[run][t, t], want[t, t])
