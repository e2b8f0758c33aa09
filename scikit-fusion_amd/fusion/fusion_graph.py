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
# Stated here as the OUTCOME the reference's masked-array statements (fusion_graph.py:464-501) have under NumPy, which
# tests/golden/fill_strategies.npz pins bit for bit (incl. non-finite values under the mask, rows / columns unknown
# throughout, both infinities in one line):
#   unknown entry      = NaN, +-inf, or masked
#   a mean             = numpy.nanmean: over the entries that are neither NaN nor masked -- an infinity COUNTS, so a line
#                        holding one has an infinite (or NaN) mean, exactly as in the reference
#   'mean', constant   -> the value is written into every unknown entry, also beneath the mask; the mask itself stays (Dfmc
#                        turns it into its completion mask, dfmc.py:77-82)
#   'row_mean'         -> every unknown entry takes the mean of its row, and the result carries no mask any more; a row
#                        without a usable mean (masked input: non-finite; plain input: NaN) takes the matrix mean
#   'col_mean'         -> the same along the columns
#   a MaskedArray that masks nothing ('row_mean' / 'col_mean'): numpy.ma masks an INFINITE line mean by itself, and the
#                        entries that would take it come back masked with 1 beneath; a NaN line mean takes the matrix
#                        mean (golden tags nomask / nomaskfinite; tools/fuzz_fill.py compares 78 000 random inputs with the
#                        reference itself)
# The same rules run on the device in skf_fill_unknown (`Relation.filled_device`).
def _values_and_mask(x):
    values = np.ma.getdata(x)
    return values, (np.ma.getmaskarray(x) if np.ma.is_masked(x) else None)


def _mean_of_known(values, hidden, axis=None):
    """numpy.nanmean with the masked entries left out: sums of the zero-substituted matrix in NumPy's own reduction order
    (the results carry the same bits as the reference's), divided by the number of entries that count."""
    left_out = np.isnan(values) if hidden is None else np.logical_or(np.isnan(values), hidden)
    total = np.where(left_out, 0.0, values).sum(axis=axis)
    count = values.size - left_out.sum() if axis is None else values.shape[axis] - left_out.sum(axis=axis)
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.true_divide(total, count)


def _unknown(values, hidden):
    bad = ~np.isfinite(values)
    return bad if hidden is None else np.logical_or(bad, hidden)


def _write_beneath(x, where, value):
    out = x.copy()
    np.ma.getdata(out)[where] = value            # (a masked entry keeps its mask: only the stored value changes)
    return out


def fill_mean(x):
    values, hidden = _values_and_mask(x)
    return _write_beneath(x, _unknown(values, hidden), _mean_of_known(values, hidden))


def fill_const(x, const):
    values, hidden = _values_and_mask(x)
    return _write_beneath(x, _unknown(values, hidden), const)


def _fill_lines(x, axis):
    """Every unknown entry <- the mean of its line (axis=1: row, axis=0: column)."""
    values, hidden = _values_and_mask(x)
    line = np.atleast_1d(_mean_of_known(values, hidden, axis=axis))
    where = _unknown(values, hidden)
    filled = np.array(values, copy=True)
    if np.ma.isMaskedArray(x) and hidden is None:
        # a MaskedArray that masks nothing: numpy.ma's division has already masked every INFINITE line mean (its domain
        # test; the data beneath becomes its fill for a division, 1), the reference's NaN test does not see those, and
        # the entries that take such a mean come out MASKED with 1 beneath; a NaN mean (a line without entries, or both
        # infinities in it) is replaced by the matrix mean, whatever that is (fusion_graph.py:483-489)
        lost = np.isinf(line)
        line = np.where(np.isnan(line), _mean_of_known(values, hidden), line)
        lost_entry = np.broadcast_to(lost[:, None] if axis == 1 else lost[None, :], values.shape) & where
        per_entry = np.broadcast_to(line[:, None] if axis == 1 else line[None, :], values.shape)
        filled[where] = per_entry[where]
        filled[lost_entry] = 1.0
        return np.ma.MaskedArray(filled, mask=lost_entry.copy())
    unusable = np.isnan(line) if hidden is None else ~np.isfinite(line)
    line = np.where(unusable, _mean_of_known(values, hidden), line)
    per_entry = np.broadcast_to(line[:, None] if axis == 1 else line[None, :], values.shape)
    filled[where] = per_entry[where]
    return filled if not np.ma.isMaskedArray(x) else np.ma.MaskedArray(filled, mask=np.zeros(values.shape, dtype=bool))


def fill_row(x):
    return _fill_lines(x, 1)


def fill_col(x):
    return _fill_lines(x, 0)


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
        if (np.ma.isMaskedArray(x) and mask is None and strategy in ('row_mean', 'col_mean')
                and np.isinf(np.ma.getdata(x)).any()):
            # a MaskedArray that masks nothing, with an infinity: numpy.ma masks the infinite line means and the
            # reference hands those entries back masked (see _fill_lines) -- a rule of numpy.ma, not of the fill; this
            # shape is imputed by the host statement and uploaded
            from .._engine import upload_matrix
            f = _FILLERS[strategy](x)
            lost = np.ma.getmaskarray(f)
            return upload_matrix(np.ma.getdata(f), dtype, runtime), (lost if lost.any() else None)
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
    # Bookkeeping rule (what reference fusion_graph.py:174-251 maintains, stated once): `relations` is the truth; an object
    # type is part of the graph exactly as long as some relation mentions it; `adjacency_matrix[row][col]` lists the
    # relations of a pair in insertion order and holds no empty lists; the two name tables follow.
    def add_relation(self, relation):
        self.relations[relation] = True
        if relation.name:
            self._name2relation[relation.name] = relation
        for ot in (relation.row_type, relation.col_type):
            self.object_types[ot] = True
            self._name2object_type[ot.name] = ot
        pairs = self.adjacency_matrix.setdefault(relation.row_type, {})
        pairs[relation.col_type] = pairs.get(relation.col_type, []) + [relation]

    def add_relations_from(self, relations):
        for relation in relations:
            self.add_relation(relation)

    def _mentioned(self, object_type):
        return any(object_type in relation for relation in self.relations)

    def _forget_type(self, object_type):
        self.adjacency_matrix.pop(object_type, None)
        for pairs in self.adjacency_matrix.values():
            pairs.pop(object_type, None)
        self._name2object_type.pop(object_type.name, None)
        self.object_types.pop(object_type, None)

    def remove_relation(self, relation):
        pairs = self.adjacency_matrix[relation.row_type]
        pairs[relation.col_type].remove(relation)
        if not pairs[relation.col_type]:
            del pairs[relation.col_type]
        del self.relations[relation]
        if relation.name:
            self._name2relation.pop(relation.name, None)
        for ot in (relation.row_type, relation.col_type):      # a type nobody mentions any more leaves with its last relation
            if ot in self.object_types and not self._mentioned(ot):
                self._forget_type(ot)

    def remove_relations_from(self, relations):
        for relation in relations:
            self.remove_relation(relation)

    def remove_object_type(self, object_type):
        for relation in [r for r in self.relations if object_type in r]:
            if relation in self.relations:
                self.remove_relation(relation)
        self._forget_type(object_type)

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
