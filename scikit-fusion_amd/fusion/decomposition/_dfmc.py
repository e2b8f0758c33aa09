"""Functional solver seam of DFMC -- same signature as the reference's ``dfmc()``
(_dfmc.py:181-184).  ``M[(i,j)]`` is a list (parallel to ``R[(i,j)]``) of boolean masks or
``None``; masked entries are unknown and are re-estimated from the current model in every
iteration (_dfmc.py:287-292, :319-325).  The relation matrices passed in are never modified
(the engine completes a device-side copy)."""
from ... import _native as nat
from ._dfmf import run_fit, run_fit_sharded, run_fit_rows, run_fit_owned


def dfmc(R, M, Theta, obj_types, obj_type2rank, max_iter=10, init_type="random_vcol",
         stopping=None, stopping_system=None, verbose=0, compute_err=False, callback=None,
         random_state=None, n_jobs=1, dtype='f64', G0=None, engine=None, shard=None):
    """Data fusion by matrix completion -- drop-in for reference ``dfmc`` (_dfmc.py:181)."""
    if shard in ('relations', 'rows', 'owned'):
        fit = {'relations': run_fit_sharded, 'rows': run_fit_rows, 'owned': run_fit_owned}[shard]
        return fit(nat.SKF_DFMC, R, M, Theta, obj_types, obj_type2rank, max_iter,
                   init_type, random_state, dtype, G0, engine, stopping, stopping_system, compute_err, callback)
    return run_fit(nat.SKF_DFMC, R, M, Theta, obj_types, obj_type2rank, max_iter, init_type,
                   stopping, stopping_system, verbose, compute_err, callback, random_state,
                   dtype, G0, engine)
