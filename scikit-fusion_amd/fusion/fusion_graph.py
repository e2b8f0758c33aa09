"""Graph data model: object types, relations (data sets) and the fusion graph.

Host-side mirror of reference ``skfusion/fusion/base/fusion_graph.py`` without the drawing
helpers (:51-172, visualisation is out of scope): ``ObjectType`` (:436-461), ``Relation``
(:513-567), fill strategies (:464-510) and the ``FusionGraph`` bookkeeping (:16-433).
Filling / preprocessing runs once per ``fuse()`` on the host in NumPy; only the resulting
matrices enter the device engine.
"""
from collections import OrderedDict
from numbers import Number
from uuid import uuid1

import numpy as np

from .base import DataFusionError

__all__ = ['FusionGraph', 'Relation', 'ObjectType']


class ObjectType(object):
    """A kind of object (genes, users, ...) with its factorisation rank.  Identity is the
    name: two ObjectType objects with equal names are the same node of the graph."""

    def __init__(self, name, rank=5):
        self.name = name
        self.rank = rank

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return isinstance(other, ObjectType) and other.name == self.name

    def __ne__(self, other):
        return not self.__eq__(other)

    def __str__(self):
        return self.name

    def __repr__(self):
        return 'ObjectType("%s")' % self.name


# ---- fill strategies: what Relation.filled() does to NaN / inf / masked entries ------------
# The operation sequences follow reference fusion_graph.py:464-501 one to one, because the
# resulting masked-ness (which Dfmc turns into its completion mask, dfmc.py:77-82) depends on
# NumPy's masked-array assignment rules; tests/golden/fill_strategies.npz pins the outcome.
def _unknown(x):
    bad = ~np.isfinite(x)
    return np.logical_or(bad, x.mask) if np.ma.is_masked(x) else bad


def fill_mean(x):
    value = np.nanmean(x)
    where = _unknown(x)
    out = x.copy()
    out[where] = value
    return out


def fill_row(x):
    per_row = np.nanmean(x, 1)
    overall = np.nanmean(x)
    if np.ma.is_masked(x):
        per_row = np.ma.filled(np.ma.masked_invalid(per_row), overall)
        where = np.logical_or(~np.isfinite(x.data), x.mask)
    else:
        per_row[np.isnan(per_row)] = overall
        where = ~np.isfinite(x)
    out = x.copy()
    out[where] = np.take(per_row, where.nonzero()[0])
    return out


def fill_col(x):
    return fill_row(x.T).T


def fill_const(x, const):
    out = x.copy()
    out[~np.isfinite(x)] = const
    if np.ma.is_masked(x):
        out.data[x.mask] = const
    return out


_FILLERS = {'mean': fill_mean, 'row_mean': fill_row, 'col_mean': fill_col}


class Relation(object):
    """A data matrix relating ``row_type`` objects (rows) to ``col_type`` objects (columns).

    ``fill_value``: 'mean' | 'row_mean' | 'col_mean' | number -- how unknown entries are
    imputed before factorisation; ``preprocessor`` / ``postprocessor``: optional callables
    applied to the filled data before fusion / to the reconstruction in ``complete()``.
    Extra keyword arguments become attributes.
    """

    def __init__(self, data, row_type, col_type, name='', row_names=None, col_names=None,
                 fill_value='mean', row_metadata=None, col_metadata=None,
                 preprocessor=None, postprocessor=None, **kwargs):
        self.data = data
        self.row_type = row_type
        self.col_type = col_type
        self.name = name
        self.row_names = row_names
        self.col_names = col_names
        self.fill_value = fill_value
        self.row_metadata = row_metadata
        self.col_metadata = col_metadata
        self.preprocessor = preprocessor
        self.postprocessor = postprocessor
        for key, value in kwargs.items():
            setattr(self, key, value)
        self._id = name or uuid1()

    def filled(self):
        if isinstance(self.fill_value, Number):
            return fill_const(self.data, self.fill_value)
        return _FILLERS[self.fill_value](self.data)

    def filled_device(self, dtype='f64', runtime=None):
        """``filled()`` on the device: the raw matrix (and its mask) are uploaded once and the unknown entries are
        imputed in HBM (``skf_fill_unknown``); returns ``(DeviceMatrix in the engine dtype, mask or None)`` where
        the mask is what ``filled()`` would leave on the result -- kept by 'mean' and constants, dropped by
        'row_mean' / 'col_mean' (reference fusion_graph.py:475-489 under NumPy's masked-assignment rules)."""
        from .._engine import fill_unknown_device
        x = self.data
        mask = np.ma.getmaskarray(x) if np.ma.isMaskedArray(x) and np.ma.is_masked(x) else None
        if isinstance(self.fill_value, Number):
            strategy, value = 'const', float(self.fill_value)
        else:
            if self.fill_value not in _FILLERS:
                raise KeyError(self.fill_value)
            strategy, value = self.fill_value, 0.0
        dm = fill_unknown_device(np.ma.getdata(x), mask, strategy, value, dtype, runtime)
        keeps_mask = strategy in ('mean', 'const')
        return dm, (mask if (mask is not None and keeps_mask) else None)

    def __contains__(self, obj_type):
        return obj_type == self.row_type or obj_type == self.col_type

    def __hash__(self):
        return hash(self._describe(str))

    def __eq__(self, other):
        return isinstance(other, Relation) and self._id == other._id

    def __ne__(self, other):
        return not self.__eq__(other)

    def _describe(self, fmt):
        middle = '"%s"' % self.name if self.name else u"→"
        return "Relation(%s %s %s)" % (fmt(self.row_type), middle, fmt(self.col_type))

    def __str__(self):
        return self._describe(str)

    def __repr__(self):
        return self._describe(repr)


class FusionGraph(object):
    """Container of relations and object types.

    ``adjacency_matrix[row_type][col_type]`` is the list of relations row_type -> col_type
    (several relations between one pair are allowed; a relation with row_type == col_type
    is a constraint); ``relations`` and ``object_types`` are insertion-ordered.
    """

    def __init__(self, relations=()):
        self.adjacency_matrix = {}
        self.relations = OrderedDict()
        self.object_types = OrderedDict()
        self._name2relation = {}
        self._name2object_type = {}
        self.add_relations_from(relations)

    n_relations = property(lambda self: len(self.relations))
    n_object_types = property(lambda self: len(self.object_types))

    def __getitem__(self, key):
        if key in self.adjacency_matrix:
            return self.adjacency_matrix[key]
        return self._name2relation.get(key)

    def __setitem__(self, key, value):
        self.adjacency_matrix[key] = value

    # ---- visualisation is not part of this engine ------------------------------------------
    def draw_graphviz(self, *args, **kwargs):
        raise NotImplementedError("graph drawing is outside the scope of skfusion_amd")

    draw_networkx = draw_graphviz

    # ---- mutation ---------------------------------------------------------------------------
    def add_relation(self, relation):
        self.relations[relation] = True
        if relation.name:
            self._name2relation[relation.name] = relation
        for ot in (relation.row_type, relation.col_type):
            self.object_types[ot] = True
            self._name2object_type[ot.name] = ot
        row = self.adjacency_matrix.setdefault(relation.row_type, {})
        row[relation.col_type] = row.get(relation.col_type, []) + [relation]

    def add_relations_from(self, relations):
        for relation in relations:
            self.add_relation(relation)

    def _isolated(self, object_type):
        return not any(True for _ in self.in_neighbors(object_type)) and \
            not any(True for _ in self.out_neighbors(object_type))

    def remove_relation(self, relation):
        row = self.adjacency_matrix[relation.row_type]
        row[relation.col_type].remove(relation)
        self.relations.pop(relation)
        if relation.name:
            self._name2relation.pop(relation.name, None)
        if not row[relation.col_type]:
            row.pop(relation.col_type, None)
        if self._isolated(relation.row_type):
            self.remove_object_type(relation.row_type)
            if relation.row_type == relation.col_type:
                return
        if self._isolated(relation.col_type):
            self.remove_object_type(relation.col_type)

    def remove_relations_from(self, relations):
        for relation in relations:
            self.remove_relation(relation)

    def remove_object_type(self, object_type):
        for relation in list(self.relations):
            if object_type in relation and relation in self.relations:
                self.remove_relation(relation)
        if object_type not in self.object_types:
            return          # dropped as a side effect of removing its last relation
        self.adjacency_matrix.pop(object_type, None)
        for row in self.adjacency_matrix.values():
            row.pop(object_type, None)
        self._name2object_type.pop(object_type.name, None)
        self.object_types.pop(object_type)

    def remove_object_types_from(self, object_types):
        for object_type in object_types:
            self.remove_object_type(object_type)

    # ---- lookup -----------------------------------------------------------------------------
    def get_relation(self, name):
        if name not in self._name2relation:
            raise DataFusionError("Relation name unknown")
        return self._name2relation[name]

    def get_relations(self, row_type, col_type):
        if row_type not in self.object_types or col_type not in self.object_types:
            raise DataFusionError("Object types are not recognized.")
        return iter(self.adjacency_matrix.get(row_type, {}).get(col_type, []))

    def get_object_type(self, name):
        if name not in self._name2object_type:
            raise DataFusionError("Object type name unknown")
        return self._name2object_type[name]

    def _resolve(self, object_type):
        return self.get_object_type(object_type) if isinstance(object_type, str) else object_type

    def get_names(self, object_type):
        """Row/column names recorded for the objects of this type, else '0', '1', ..."""
        object_type = self._resolve(object_type)
        size = 0
        for rel in self.out_relations(object_type):
            if rel.row_names:
                return rel.row_names
            size = rel.data.shape[0]
        for rel in self.in_relations(object_type):
            if rel.col_names:
                return rel.col_names
            size = rel.data.shape[1]
        return [str(k) for k in range(size)]

    def get_metadata(self, object_type):
        """Per-object metadata dictionaries merged over all relations of the type."""
        object_type = self._resolve(object_type)
        merged = [{} for _ in self.get_names(object_type)]
        for rel in self.out_relations(object_type):
            for dst, src in zip(merged, rel.row_metadata or ()):
                dst.update(src)
        for rel in self.in_relations(object_type):
            for dst, src in zip(merged, rel.col_metadata or ()):
                dst.update(src)
        return merged

    def _known(self, object_type):
        if object_type not in self.object_types:
            raise DataFusionError("Object type not in the fusion graph.")

    def out_relations(self, object_type):
        self._known(object_type)
        for rels in list(self.adjacency_matrix.get(object_type, {}).values()):
            for relation in rels:
                yield relation

    def in_relations(self, object_type):
        self._known(object_type)
        for row in list(self.adjacency_matrix.values()):
            for relation in row.get(object_type, ()):
                yield relation

    def out_neighbors(self, object_type):
        self._known(object_type)
        return iter(list(self.adjacency_matrix.get(object_type, {}).keys()))

    def in_neighbors(self, object_type):
        self._known(object_type)
        for row_type, row in list(self.adjacency_matrix.items()):
            if row.get(object_type):
                yield row_type

    def __str__(self):
        return "FusionGraph(Object types: %d, Relations: %d)" % (
            len(self.object_types), len(self.relations))

    def __repr__(self):
        return "FusionGraph(Object types=%r, Relations=%r)" % (
            list(self.object_types.keys()), list(self.relations.keys()))
