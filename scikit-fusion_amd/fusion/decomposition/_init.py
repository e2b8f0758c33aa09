"""Factor initialisers (host side, NumPy).

They run once per fit, consume a ``numpy.random.RandomState`` (MT19937) stream and must draw
from it in exactly the reference's order for seeded results to agree (reference
``_init.py:6-61``: `random` :11-17, `random_c` :20-41, `random_vcol` :44-61), so they stay
on the host; the resulting G0 is uploaded once.  ``R`` maps (row_type, col_type) to the FIRST
relation matrix of that pair (reference _dfmf.py:191).
"""
import numpy as np


def _views_of(obj_type, R):
    """Every relation that touches obj_type, oriented so that its rows are obj_type's objects, with a key that names the
    oriented view ((pair, transposed?))."""
    for pair, mat in R.items():
        if obj_type in pair:
            yield (mat if obj_type == pair[0] else mat.T), (pair, obj_type != pair[0])


def _random(obj_types, n_obj, rank, R, random_state):
    return {(t, t): random_state.rand(n_obj[t], rank[t]) for t in _ordered(obj_types)}


def _ordered(obj_types):
    # iteration order of the caller's container == RNG consumption order
    return list(obj_types)


def _column_means_init(obj_types, n_obj, rank, R, random_state, pool_of):
    """G_t = 1e-5 + sum over relations of |mean of p randomly chosen columns| per factor column.
    ``pool_of(view)`` returns the column index pool that is shuffled before every draw."""
    G = {}
    for t in _ordered(obj_types):
        c = rank[t]
        acc = np.full((n_obj[t], c), 1e-5)
        for view, key in _views_of(t, R):
            n_cols = view.shape[1]
            take = int(.2 * n_cols)
            pool = pool_of(view, key)
            part = np.zeros((n_obj[t], c))
            for k in range(c):
                random_state.shuffle(pool)
                part[:, k] = view[:, pool[:take]].mean(axis=1)
            acc += np.abs(part)
        G[t, t] = acc
    return G


def _random_vcol(obj_types, n_obj, rank, R, random_state):
    return _column_means_init(obj_types, n_obj, rank, R, random_state,
                              lambda view, key: np.arange(view.shape[1]))


def _random_c(obj_types, n_obj, rank, R, random_state, pools=None):
    # `pools`: {(type pair of the relation, transposed?): the strongest half of its columns} kept by the caller across the
    # restarts of ONE fit -- the ranking by column norm draws nothing from the random stream and the matrices do not change
    # between restarts, yet at n_run = 10 on the README graph it was 9 of the 17 ms the ten initialisations took (3 800
    # norm calls).  Every call still starts from a FRESH copy of the ranked list, as the reference builds it anew.
    def strongest_half(view, key):
        if pools is not None and key in pools:
            return list(pools[key])
        n_cols = view.shape[1]
        norms = [np.linalg.norm(view[:, k], 2) for k in range(n_cols)]
        order = sorted(range(n_cols), key=norms.__getitem__, reverse=True)   # stable, descending
        half = order[:int(.5 * n_cols)]
        if pools is not None:
            pools[key] = tuple(half)
        return half
    return _column_means_init(obj_types, n_obj, rank, R, random_state, strongest_half)


INIT_TYPES = {"random": _random, "random_c": _random_c, "random_vcol": _random_vcol}


def initialize(obj_types, obj_type2n_obj, obj_type2rank, R, init_typ, random_state, pools=None):
    """Unknown ``init_typ`` raises KeyError like the reference dispatcher (_init.py:7-8).  `pools` (optional, a dict the
    caller keeps for the restarts of one fit): see `_random_c`."""
    if init_typ == 'random_c' and pools is not None:
        return _random_c(obj_types, obj_type2n_obj, obj_type2rank, R, random_state, pools)
    return INIT_TYPES[init_typ](obj_types, obj_type2n_obj, obj_type2rank, R, random_state)
