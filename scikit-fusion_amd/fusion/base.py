"""Result accessors shared by the fusers (host side, pure Python).

Mirrors the behaviour of reference ``skfusion/fusion/base/base.py``:
``factor`` (:35-56), ``backbone`` (:169-189), ``complete`` (:119-146), ``chain`` (:69-96),
``FusionTransform._validate_graph`` (:224-231) and ``DataFusionError`` (:250), including the
"generator when n_run > 1 and run is None" convention.
"""
from collections import defaultdict

import numpy as np

__all__ = ['FusionBase', 'FusionFit', 'FusionTransform', 'DataFusionError', 'save_fit', 'load_fit']


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

    # ---- chained latent profiles on the device --------------------------------------------------------
    # The reference's base.py:69-96 only enumerates the paths; what its users compute from them is, per path c,
    #     bb = reduce(np.dot, [backbone(first relation c[i] -> c[i+1])])          (examples/dicty_chaining.py:43-45)
    #     profile = G_row . bb . G_col^T        (dicty_chaining.py:46-49)    or    G_row . bb   (pharma_chaining.py:48-50)
    # hstack-ed over the object types and their paths (the path [row_type] contributes G_row itself).  At BASELINE config-3
    # sizes one such block is 50k x 100k: it is produced blockwise on the device like `complete_blocks`.
    def _chain_source(self, row_type, run):
        """(the fit whose backbones / column factors the paths use, the factor of the starting type in run `run`)"""
        raise NotImplementedError

    def chain_paths(self, row_type, col_types=None, skip=()):
        """[(col_type, path)] in the order the reference examples walk them: object types in `col_types` order (default:
        the graph's), the paths of `chain()` for each, types in `skip` left out."""
        fit, _ = self._chain_source(row_type, 0)
        cols = list(fit.fusion_graph.object_types) if col_types is None else list(col_types)
        out = []
        for ct in cols:
            if any(ct is s or ct == s for s in skip):
                continue
            for path in FusionBase.chain(fit, row_type, ct):      # the FITTED model's graph, also for a transformer
                out.append((ct, path))
        return out

    def chain_backbone(self, path, run=None):
        """Product of the backbones of the FIRST relation of every hop of `path` (as the examples take them), reduced on
        the device in f64 (`_engine.chain_backbone`); None for the one-type path."""
        from .._engine import chain_backbone
        run = 0 if run is None else run
        fit, _ = self._chain_source(path[0], run)
        hops = []
        for a, b in zip(path[:-1], path[1:]):
            rels = list(fit.fusion_graph.get_relations(a, b))
            if not rels:
                raise DataFusionError("No relation %s -> %s on the path" % (a.name, b.name))
            hops.append(fit.backbone(rels[0], run))
        return chain_backbone(hops) if hops else None

    def chain_profile_blocks(self, row_type, col_types=None, block_rows=4096, run=None, dtype='f64', project=True,
                             skip=()):
        """Yields ``(row_slice, X[row_slice])``: the chained latent profile of the objects of ``row_type`` (for a
        transformer: of its target's NEW objects), at most ``block_rows`` rows at a time -- for every path of
        ``chain_paths`` the block ``G_row[rows] . (prod of backbones) . G_col^T`` (``project=True``, reference
        examples/dicty_chaining.py:40-53) or ``G_row[rows] . (prod of backbones)`` (``project=False``,
        pharma_chaining.py:43-53), the one-type path contributing ``G_row[rows]`` itself, hstack-ed in path order.  The
        backbone products (f64) and the column factors are uploaded once and stay resident across the blocks; the two
        GEMMs of a block run in the master type of ``dtype`` ('f64' | 'f32' | 'bf16' -> f32).  float64 ndarrays."""
        from .._engine import DeviceReconstructor
        run = 0 if run is None else run
        fit, G_row = self._chain_source(row_type, run)
        for who in {id(self): self, id(fit): fit}.values():          # (a transformer and the fitted model it folds into)
            if not 0 <= int(run) < int(getattr(who, 'n_run', 1)):
                raise DataFusionError("run %d requested, %s holds %d" % (run, type(who).__name__, getattr(who, 'n_run', 1)))
        G_row = np.asarray(G_row)
        parts, col_on_device = [], {}             # one uploaded copy of a column factor, however many paths end in its type
        for ct, path in self.chain_paths(row_type, col_types, skip):
            bb = self.chain_backbone(path, run)
            if bb is None:
                parts.append(None)
            else:
                rec = DeviceReconstructor(bb, fit.factor(ct, run) if project else None, dtype=dtype,
                                          G_col_device=col_on_device.get(ct))
                if project:
                    col_on_device.setdefault(ct, rec.b)
                parts.append(rec)
        if not parts:
            raise DataFusionError("No path from %s to the requested object types" % row_type.name)
        for r0 in range(0, G_row.shape[0], int(block_rows)):
            sl = slice(r0, min(r0 + int(block_rows), G_row.shape[0]))
            rows = G_row[sl]
            yield sl, np.hstack([np.asarray(rows, dtype=np.float64) if rec is None else rec.block(rows) for rec in parts])

    def chain_profile(self, row_type, col_types=None, run=None, dtype='f64', project=True, skip=(), block_rows=4096):
        """The whole profile matrix (``np.vstack`` of ``chain_profile_blocks``) -- for sizes that fit the host."""
        return np.vstack([blk for _, blk in self.chain_profile_blocks(row_type, col_types, block_rows, run, dtype,
                                                                      project, skip)])

    def __repr__(self):
        inner = ', '.join('{}={}'.format(k, v) for k, v in (self._params or {}).items())
        return '{}({})'.format(type(self).__name__, inner)

    __str__ = __repr__


class FusionFit(FusionBase):
    """Accessors of a fitted fuser: backbones and reconstructed relations."""

    def _chain_source(self, row_type, run):
        return self, self.factor(row_type, run)

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


    def save(self, path):
        """Write the fitted model to ``path`` (one ``.npz``; see ``save_fit``)."""
        return save_fit(self, path)

    @staticmethod
    def load(path, fusion_graph=None):
        """Read a model written by ``save`` (see ``load_fit``)."""
        return load_fit(path, fusion_graph)

    def complete_blocks(self, relation, block_rows=4096, run=None, dtype='f64', device=False):
        """Blockwise reconstruction on the device: yields ``(row_slice, R_hat[row_slice])`` with
        at most ``block_rows`` rows per block, so that a relation whose dense reconstruction is too
        large for the host (BASELINE config 3: up to 40 GB) can be consumed piece by piece.  The backbone and the
        column factor are uploaded once and stay resident in HBM across the blocks.  ``device=True`` yields the
        block as a device-resident matrix in the engine's MASTER type (f64 for ``dtype='f64'``, f32 for ``'f32'`` and
        ``'bf16'``) that aliases a scratch buffer: it is valid until the next block and is NOT post-processed -- a
        relation with a postprocessor is refused with ``device=True``; otherwise a float64 ndarray, post-processed
        per block (exact for element-wise postprocessors).
        Extension of the reference API (``complete`` itself is unchanged)."""
        from .._engine import DeviceReconstructor
        if device and relation.postprocessor:
            raise DataFusionError("complete_blocks(device=True) yields raw device blocks; relation %s -> %s has a "
                                  "postprocessor (use device=False, or apply it to the blocks yourself)"
                                  % (relation.row_type.name, relation.col_type.name))
        run = 0 if run is None else run
        G1 = self.factor(relation.row_type, run)
        rec = DeviceReconstructor(self.backbone(relation, run), self.factor(relation.col_type, run), dtype=dtype)
        for r0 in range(0, G1.shape[0], int(block_rows)):
            sl = slice(r0, min(r0 + int(block_rows), G1.shape[0]))
            block = rec.block(G1[sl], device=device)
            if relation.postprocessor and not device:
                block = relation.postprocessor(block)
            yield sl, block


# ---- persistence of a fitted model (the reference has none: its accessors base.py:35-56,
# 169-189 are the only consumers of factors_ / backbones_) ----------------------------------------------
_FORMAT = 'skfusion_amd.fit/1'


def _plain(value):
    """Constructor parameters that survive JSON; callables, RandomState objects ... are dropped (None)."""
    if value is None or isinstance(value, (bool, int, float, str)):
        return value
    if isinstance(value, (np.integer, np.floating)):
        return value.item()
    if isinstance(value, (list, tuple)):
        return [_plain(v) for v in value]
    return None


def save_fit(fuser, path):
    """Fitted Dfmf / Dfmc -> ONE ``.npz`` file: every factor G (per object type and run) and backbone S (per
    relation and run) as float64 arrays ``G/<type index>/<run>``, ``S/<relation index>/<run>``, plus a JSON
    header (format tag, class, constructor parameters, object types with rank and size, relations by row / column
    type and position among the relations of that pair)."""
    import json
    graph = fuser.fusion_graph
    types = list(graph.object_types)
    rels = list(graph.relations)
    n_run = int(fuser.n_run)
    arrays = {}
    for a, ot in enumerate(types):
        runs = fuser.factors_[ot]
        if len(runs) != n_run:
            raise DataFusionError("Object type %s has %d fitted factors, expected %d" % (ot.name, len(runs), n_run))
        for k, G in enumerate(runs):
            arrays['G/%d/%d' % (a, k)] = np.asarray(G, dtype=np.float64)
    seen = {}
    rel_meta = []
    for b, rel in enumerate(rels):
        pair = (rel.row_type.name, rel.col_type.name)
        pos = seen.get(pair, 0)
        seen[pair] = pos + 1
        rel_meta.append({'row': pair[0], 'col': pair[1], 'name': str(getattr(rel, 'name', '') or ''), 'position': pos,
                         'shape': [int(d) for d in np.shape(rel.data)]})
        # a same-type relation is a constraint (Theta): it has no backbone.  `.get`: no entry is inserted into the
        # defaultdict, so `backbone(theta_relation)` keeps raising "Unknown relation." after a save
        runs = fuser.backbones_.get(rel, [])
        rel_meta[-1]['n_backbones'] = len(runs)
        for k, S in enumerate(runs):
            arrays['S/%d/%d' % (b, k)] = np.asarray(S, dtype=np.float64)
    meta = {'format': _FORMAT, 'class': type(fuser).__name__, 'n_run': n_run,
            'params': {k: _plain(v) for k, v in (fuser._params or {}).items()},
            'object_types': [{'name': str(ot.name), 'rank': int(ot.rank),
                              'n_objects': int(np.shape(fuser.factors_[ot][0])[0])} for ot in types],
            'relations': rel_meta}
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode('utf-8'), dtype=np.uint8)
    with open(path, 'wb') as fh:
        np.savez_compressed(fh, **arrays)
    return path


def load_fit(path, fusion_graph=None):
    """Inverse of ``save_fit``: a fuser of the saved class whose ``factor`` / ``backbone`` / ``complete`` /
    ``chain`` accessors work (including the generator convention for ``n_run > 1``).  With ``fusion_graph`` the
    factors are attached to ITS object types / relations (matched by name and by position among the relations of
    a type pair; sizes are checked); without one a skeleton graph of the saved types and relation shapes is
    built (relation data: read-only NaN views that take no memory)."""
    import json
    from .fusion_graph import FusionGraph, ObjectType, Relation
    with np.load(path, allow_pickle=False) as z:
        meta = json.loads(bytes(z['meta'].tobytes()).decode('utf-8'))
        if meta.get('format') != _FORMAT:
            raise DataFusionError("Not a saved fusion model: %r" % (meta.get('format'),))
        arrays = {k: z[k] for k in z.files if k != 'meta'}
    if fusion_graph is None:
        by_name = {t['name']: ObjectType(t['name'], t['rank']) for t in meta['object_types']}
        fusion_graph = FusionGraph([Relation(np.broadcast_to(np.nan, tuple(r['shape'])), by_name[r['row']],
                                             by_name[r['col']], name=r['name']) for r in meta['relations']])
    from . import decomposition
    cls = getattr(decomposition, meta['class'], None)
    if cls is None or not issubclass(cls, FusionFit):
        raise DataFusionError("Unknown fuser class %r" % (meta['class'],))
    params = {k: v for k, v in meta['params'].items() if v is not None}
    fuser = cls(**params)
    fuser.fusion_graph = fusion_graph
    n_run = int(meta['n_run'])
    names = {ot.name: ot for ot in fusion_graph.object_types}
    for a, t in enumerate(meta['object_types']):
        if t['name'] not in names:
            raise DataFusionError("Object type %s is not included in the fusion scheme" % t['name'])
        ot = names[t['name']]
        for k in range(n_run):
            G = arrays['G/%d/%d' % (a, k)]
            if G.shape != (t['n_objects'], t['rank']):
                raise DataFusionError("Factor of %s has shape %r" % (t['name'], G.shape))
            fuser.factors_[ot].append(G)
    for b, r in enumerate(meta['relations']):
        cands = list(fusion_graph.get_relations(names[r['row']], names[r['col']]))
        if r['position'] >= len(cands):
            raise DataFusionError("Relation %s -> %s #%d is not in the fusion graph" % (r['row'], r['col'], r['position']))
        rel = cands[r['position']]
        if list(np.shape(rel.data)) != list(r['shape']):
            raise DataFusionError("Relation %s -> %s: data shape %r, saved %r"
                                  % (r['row'], r['col'], np.shape(rel.data), r['shape']))
        have = int(r.get('n_backbones', 0 if r['row'] == r['col'] else n_run))
        for k in range(have):              # (constraints were saved without backbones)
            fuser.backbones_[rel].append(arrays['S/%d/%d' % (b, k)])
    return fuser


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

    def _chain_source(self, row_type, run):
        # the paths, backbones and column factors are the fitted model's; the rows are the target's NEW objects
        # (reference examples: profile(fuser, transformer) takes transformer.factor(gene) with fuser.chain / backbone)
        if row_type is not None and row_type is not self.target:
            raise DataFusionError("Starting type should be target type: %s" % self.target.name)
        return self.fuser, self.factor(self.target, run)

    def chain(self, row_type=None, col_type=None):
        if row_type is not None and col_type is not None and row_type is not self.target:
            raise DataFusionError("Starting type should be target type: %s" % self.target.name)
        if col_type is None:
            col_type = row_type
        return FusionBase.chain(self, self.target, col_type)
