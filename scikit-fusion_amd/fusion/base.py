"""Result accessors shared by the fusers (host side, pure Python).

Mirrors the behaviour of reference ``skfusion/fusion/base/base.py``:
``factor`` (:35-56), ``backbone`` (:169-189), ``complete`` (:119-146), ``chain`` (:69-96),
``FusionTransform._validate_graph`` (:224-231) and ``DataFusionError`` (:250), including the
"generator when n_run > 1 and run is None" convention.
"""
from collections import defaultdict

import numpy as np

__all__ = ['FusionBase', 'FusionFit', 'FusionTransform', 'DataFusionError']


class DataFusionError(Exception):
    """API misuse: unknown object type / relation, inconsistent graph."""


class FusionBase(object):
    """Holds the constructor parameters and the fitted factors of all runs.

    ``factors_[object_type][run]`` and ``backbones_[relation][run]`` are filled by the
    concrete fusers.
    """
    _params = None

    def __init__(self):
        self.factors_ = defaultdict(list)
        self.backbones_ = defaultdict(list)

    def _set_params(self, values):
        # `values` is the vars() of the subclass constructor (reference base.py:29-33)
        self._params = {k: v for k, v in values.items() if k not in ('self', '__class__')}
        self.__dict__.update(self._params)

    # one run -> the matrix; several runs and no explicit run -> an iterator over runs
    def _select(self, per_run, run):
        if self.n_run > 1 and run is None:
            return (per_run[k] for k in range(self.n_run))
        return per_run[0 if run is None else run]

    def factor(self, object_type, run=None):
        """Latent factor G of ``object_type`` (n_objects x rank)."""
        if object_type not in self.fusion_graph.object_types:
            raise DataFusionError("Object type %s is not included in the fusion scheme"
                                  % object_type.name)
        if object_type not in self.factors_:
            raise DataFusionError("Unknown object type.")
        return self._select(self.factors_[object_type], run)

    def chain(self, row_type, col_type):
        """All simple paths row_type -> ... -> col_type along relation directions
        (breadth first, shortest first), as lists of object types."""
        frontier = [[row_type]]
        if row_type == col_type:
            yield frontier[0]
        while frontier:
            nxt = []
            for path in frontier:
                for ot in self.fusion_graph.out_neighbors(path[-1]):
                    if ot in path:
                        continue
                    longer = path + [ot]
                    if ot == col_type:
                        yield longer
                    else:
                        nxt.append(longer)
            frontier = nxt

    def __repr__(self):
        inner = ', '.join('{}={}'.format(k, v) for k, v in (self._params or {}).items())
        return '{}({})'.format(type(self).__name__, inner)

    __str__ = __repr__


class FusionFit(FusionBase):
    """Accessors of a fitted fuser: backbones and reconstructed relations."""

    def backbone(self, relation, run=None):
        """Backbone S of ``relation`` (rank_row x rank_col)."""
        types = self.fusion_graph.object_types
        if relation.row_type not in types or relation.col_type not in types:
            raise DataFusionError('Object types are not recognized.')
        if relation not in self.backbones_:
            raise DataFusionError("Unknown relation.")
        return self._select(self.backbones_[relation], run)

    def _reconstruct(self, relation, run):
        G1 = self.factor(relation.row_type, run)
        S12 = self.backbone(relation, run)
        G2 = self.factor(relation.col_type, run)
        approx = np.dot(G1, np.dot(S12, G2.T))
        if relation.postprocessor:
            approx = relation.postprocessor(approx)
        return approx

    def complete(self, relation, run=None):
        """Reconstructed relation ``G_row S G_col^T`` (post-processed if the relation has a
        postprocessor)."""
        types = self.fusion_graph.object_types
        if relation.row_type not in types or relation.col_type not in types:
            raise DataFusionError("Object type %s or %s are not included in the fusion scheme"
                                  % (relation.row_type.name, relation.col_type.name))
        if self.n_run > 1 and run is None:
            return (self._reconstruct(relation, k) for k in range(self.n_run))
        return self._reconstruct(relation, 0 if run is None else run)


    def complete_blocks(self, relation, block_rows=4096, run=None, dtype='f64'):
        """Blockwise reconstruction on the device: yields ``(row_slice, R_hat[row_slice])`` with
        at most ``block_rows`` rows per block, so that a relation whose dense reconstruction is too
        large for the host (BASELINE config 3: up to 40 GB) can be consumed piece by piece.
        A postprocessor is applied per block (exact for element-wise postprocessors).
        Extension of the reference API (``complete`` itself is unchanged)."""
        from .._engine import device_reconstruct
        run = 0 if run is None else run
        G1 = self.factor(relation.row_type, run)
        S12 = self.backbone(relation, run)
        G2 = self.factor(relation.col_type, run)
        for r0 in range(0, G1.shape[0], int(block_rows)):
            sl = slice(r0, min(r0 + int(block_rows), G1.shape[0]))
            block = device_reconstruct(G1[sl], S12, G2, dtype=dtype)
            if relation.postprocessor:
                block = relation.postprocessor(block)
            yield sl, block


class FusionTransform(FusionBase):
    """Base of the online (fold-in) transformers: attributes ``target``, ``fusion_graph``,
    ``fuser``."""

    def _validate_graph(self):
        if self.target not in self.fusion_graph.object_types:
            raise DataFusionError("Object type %s is not included in the fusion scheme."
                                  % self.target.name)
        for relation in self.fusion_graph.relations:
            if self.target not in (relation.row_type, relation.col_type):
                raise DataFusionError("Relation must include target object type: %s."
                                      % self.target.name)

    def chain(self, row_type=None, col_type=None):
        if row_type is not None and col_type is not None and row_type is not self.target:
            raise DataFusionError("Starting type should be target type: %s" % self.target.name)
        if col_type is None:
            col_type = row_type
        return FusionBase.chain(self, self.target, col_type)
