"""``Dfmc`` -- data fusion by matrix completion, class layer of the drop-in boundary
(reference ``skfusion/fusion/decomposition/dfmc.py:18-115``).  Masked entries of a relation
(numpy masked arrays that are still masked after fill + preprocess) are treated as unknown
and completed by the model; the relation objects' data is never modified."""
import numpy as np

from ..base import FusionFit
from . import _dfmc
from ..._distributed import my_runs, gather_runs
from .dfmf import (graph_matrices, store_runs, initial_factors, _random_state, concurrent_streams,
                   device_fill_dtype)

__all__ = ['Dfmc']


class Dfmc(FusionFit):
    """Parameters identical to the reference: max_iter=100, init_type='random_c', n_run=1,
    stopping=None, stopping_system=None, verbose=0, compute_err=False, callback=None,
    random_state=None, n_jobs=1.  Addition: dtype='f64' | 'f32'."""

    def __init__(self, max_iter=100, init_type='random_c', n_run=1, stopping=None,
                 stopping_system=None, verbose=0, compute_err=False, callback=None,
                 random_state=None, n_jobs=1, dtype='f64', shard='runs', device_fill=False):
        super(Dfmc, self).__init__()
        self._set_params(vars())

    def fuse(self, fusion_graph):
        self.fusion_graph = fusion_graph
        self.random_state = _random_state(self.random_state)
        object_types = list(fusion_graph.object_types)
        rank = {ot: int(ot.rank) for ot in object_types}
        R, Theta, M = graph_matrices(fusion_graph, with_masks=True, device_dtype=device_fill_dtype(self))
        G0 = initial_factors(R, object_types, rank, self.init_type, self.random_state, self.n_run)
        kw = dict(R=R, M=M, Theta=Theta, obj_types=object_types, obj_type2rank=rank,
                  max_iter=self.max_iter, init_type=self.init_type, stopping=self.stopping,
                  stopping_system=self.stopping_system, verbose=self.verbose,
                  compute_err=self.compute_err, callback=self.callback,
                  random_state=self.random_state, n_jobs=self.n_jobs, dtype=self.dtype)
        if self.shard in ('relations', 'rows', 'owned'):
            store_runs(self, [_dfmc.dfmc(G0=G0[k], shard=self.shard, **kw) for k in range(self.n_run)])
            return self
        n_streams = concurrent_streams(self)
        if n_streams:                                   # n_jobs restarts side by side on this GPU
            from ... import _native as nat
            from ._dfmf import run_fits_concurrent
            store_runs(self, run_fits_concurrent(nat.SKF_DFMC, R, M, Theta, object_types, rank, self.max_iter,
                                                 self.dtype, G0, None, n_streams))
            return self
        local = {k: _dfmc.dfmc(G0=G0[k], **kw) for k in my_runs(self.n_run)}
        store_runs(self, gather_runs(local, self.n_run))
        return self
