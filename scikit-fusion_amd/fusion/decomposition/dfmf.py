"""``Dfmf`` and ``DfmfTransform`` -- the class layer of the drop-in boundary.

Same constructor keywords, ``fuse`` / ``transform`` entry points and result containers as the
reference (``skfusion/fusion/decomposition/dfmf.py``: Dfmf :18-106, DfmfTransform :118-204).
What changes is where the arithmetic runs: each of the ``n_run`` restarts is one device plan
in ``libskfusion_hip.so`` (the reference ships every run to a joblib worker, dfmf.py:87-95).
Engine-specific keywords (``dtype``) are additions with behaviour-preserving defaults.
"""
from collections import defaultdict
from itertools import product

import numpy as np

from ..base import FusionFit, FusionTransform
from ..._distributed import my_runs, gather_runs
from ..._engine import count_objects
from . import _dfmf
from ._init import initialize

__all__ = ['Dfmf', 'DfmfTransform']


def _random_state(obj):
    return obj if isinstance(obj, np.random.RandomState) else np.random.RandomState(obj)


def graph_matrices(fusion_graph, with_masks=False, device_dtype=None):
    """FusionGraph -> (R, Theta[, M]) dictionaries in the reference's walking order
    (dfmf.py:70-85, dfmc.py:70-93): pairs from product(object_types, repeat=2), each relation
    filled, then preprocessed; relations between two different types go to R, same-type
    relations are constraints (Theta).  For a masked result the raw ``.data`` is used and,
    with ``with_masks``, the mask is kept as the completion mask.
    ``device_dtype``: relations (not constraints) without a preprocessor are filled ON THE DEVICE and enter the
    dictionaries as device-resident matrices of that engine dtype (``Relation.filled_device``)."""
    R, Theta, M = {}, {}, {}
    for row_type, col_type in product(fusion_graph.object_types, repeat=2):
        for relation in fusion_graph.get_relations(row_type, col_type):
            mask = None
            if device_dtype and not relation.preprocessor and relation.row_type != relation.col_type:
                data, mask = relation.filled_device(device_dtype)
            else:
                data = relation.filled()
                if relation.preprocessor:
                    data = relation.preprocessor(data)
                if np.ma.is_masked(data):
                    mask = data.mask
                    data = data.data
            key = (relation.row_type, relation.col_type)
            if relation.row_type != relation.col_type:
                R.setdefault(key, []).append(data)
                M.setdefault(key, []).append(mask)
            else:
                Theta.setdefault(key, []).append(data)
    return (R, Theta, M) if with_masks else (R, Theta)


def initial_factors(R, object_types, rank, init_type, random_state, n_run):
    """G0 of every restart, drawn in run order from the ONE shared RandomState (the reference
    passes the same generator object to all runs, dfmf.py:92, consumed sequentially for
    n_jobs=1).  Every rank draws all of them so that the result of run k does not depend on how
    many GPUs share the work."""
    n_obj = count_objects(object_types, R)
    # ('random' never reads the relations: they may live on the device already, `device_fill`)
    R_first = {} if init_type == 'random' else {k: np.asarray(v[0], dtype=float) for k, v in R.items()}
    pools = {}                                   # column rankings of `random_c`, shared by the restarts (R_first stays alive here)
    return [initialize(object_types, n_obj, rank, R_first, init_type, random_state, pools)
            for _ in range(n_run)]


def device_fill_dtype(fuser):
    """The engine dtype when the fill strategies of the relations may run on the device: asked for
    (`device_fill=True`), whole relations on this process (shard='runs'), and an initialiser that does not read the
    filled values on the host ('random'; the column-mean initialisers `random_c` / `random_vcol` do)."""
    if getattr(fuser, 'device_fill', False) and fuser.shard == 'runs' and fuser.init_type == 'random':
        return fuser.dtype
    return None


def concurrent_streams(fuser):
    """`n_jobs` -> how many of the `n_run` restarts share this GPU concurrently, each on a stream of
    its own (0: one after the other).  The reference hands the restarts to `n_jobs` joblib workers
    (dfmf.py:87-95); here they stay in one process.  Only the plain loop qualifies (no callback /
    stopping / error logging, which need the host every iteration), and a process group shares
    the restarts out over the GPUs instead."""
    from ..._distributed import world
    if fuser.n_run < 2 or fuser.callback or fuser.stopping or fuser.stopping_system or fuser.compute_err:
        return 0
    if fuser.shard != 'runs' or world()[1] > 1:
        return 0
    nj = fuser.n_jobs
    if nj is None or nj in (0, 1):
        return 0
    return min(fuser.n_run, 16 if nj < 0 else int(nj), 16)


def shared_launches(fuser):
    """Several restarts, the plain loop, one GPU: candidates for skf_iterate_batch (every launch serves all restarts of a
    small graph; results identical to one restart after the other, so `n_jobs` need not ask for it)."""
    from ..._distributed import world
    if fuser.n_run < 2 or fuser.callback or fuser.stopping or fuser.stopping_system or fuser.compute_err:
        return False
    if not (fuser.shard == 'runs' and world()[1] <= 1 and fuser.dtype in ('f64', 'f32')):
        return False
    # the schedule for small graphs, decided here on the host from the library's own limits (skf_small_graph_limits) -- not
    # by uploading the graph and binding a plan only to ask skf_plan_batchable, and not from a second copy of the constants
    from ..._engine import small_graph_limits
    lim = small_graph_limits()
    graph = fuser.fusion_graph
    types = list(graph.object_types)
    if len(types) > lim['max_types'] or any(int(ot.rank) > lim['max_rank'] for ot in types):
        return False
    n_rel = n_theta = 0
    for rel in graph.relations:
        if max(rel.data.shape) > lim['max_objects']:
            return False
        if rel.row_type is rel.col_type:
            n_theta += 1
            n = rel.data.shape[0]
            nnz = int(np.count_nonzero(np.ma.getdata(rel.data)))
            if nnz == 0 or nnz > n * n // lim['constraint_nnz_divisor']:
                return False
        else:
            n_rel += 1
    return 1 <= n_rel <= lim['max_relations'] and n_theta <= lim['max_constraints']


def store_runs(fuser, runs):
    """(G, S) per run -> factors_[object_type][run], backbones_[relation][run]
    (dfmf.py:97-105)."""
    fuser.factors_ = defaultdict(list)
    fuser.backbones_ = defaultdict(list)
    graph = fuser.fusion_graph
    for G, S in runs:
        for (object_type, _), factor in G.items():
            fuser.factors_[object_type].append(factor)
        for (row_type, col_type), backbones in S.items():
            for k, relation in enumerate(graph.get_relations(row_type, col_type)):
                fuser.backbones_[relation].append(backbones[k])


class Dfmf(FusionFit):
    """Data fusion by matrix factorization.

    Parameters (identical to the reference): max_iter=100, init_type='random_c', n_run=1,
    stopping=None, stopping_system=None, verbose=0, compute_err=False, callback=None,
    random_state=None, n_jobs=1 (here: that many restarts run concurrently on streams of one GPU;
    results do not depend on it).  Additions: dtype='f64' | 'f32' | 'bf16' (device arithmetic),
    shard='runs' | 'relations' | 'rows' | 'owned' (what a torch.distributed process group shares out: whole
    restarts -- no collective; the relations of each restart -- one all-reduce per iteration;
    balanced row blocks of the relations -- all-reduces of W, Q and E / D per iteration; or the rows of
    every object type with the matching rows of its relations -- a reduce-scatter of each partial Q and an
    all-gather of the updated factor rows, the form with the least exchange).
    """

    def __init__(self, max_iter=100, init_type='random_c', n_run=1, stopping=None,
                 stopping_system=None, verbose=0, compute_err=False, callback=None,
                 random_state=None, n_jobs=1, dtype='f64', shard='runs', device_fill=False):
        super(Dfmf, self).__init__()
        self._set_params(vars())

    def fuse(self, fusion_graph):
        self.fusion_graph = fusion_graph
        self.random_state = _random_state(self.random_state)
        object_types = list(fusion_graph.object_types)
        rank = {ot: int(ot.rank) for ot in object_types}
        R, Theta = graph_matrices(fusion_graph, device_dtype=device_fill_dtype(self))
        G0 = initial_factors(R, object_types, rank, self.init_type, self.random_state, self.n_run)
        kw = dict(R=R, Theta=Theta, obj_types=object_types, obj_type2rank=rank,
                  max_iter=self.max_iter, init_type=self.init_type, stopping=self.stopping,
                  stopping_system=self.stopping_system, verbose=self.verbose,
                  compute_err=self.compute_err, callback=self.callback,
                  random_state=self.random_state, n_jobs=self.n_jobs, dtype=self.dtype)
        if self.shard in ('relations', 'rows', 'owned'):                   # all GPUs cooperate on every restart
            store_runs(self, [_dfmf.dfmf(G0=G0[k], shard=self.shard, **kw) for k in range(self.n_run)])
            return self
        if shared_launches(self):                       # restarts of a SMALL graph share their launches, whatever n_jobs says
            from ... import _native as nat
            runs = _dfmf.run_fits_concurrent(nat.SKF_DFMF, R, None, Theta, object_types, rank, self.max_iter, self.dtype,
                                             G0, None, min(self.n_run, 32), batch_only=True)
            if runs is not None:
                store_runs(self, runs)
                return self
        n_streams = concurrent_streams(self)
        if n_streams:                                   # n_jobs restarts side by side on this GPU
            from ... import _native as nat
            store_runs(self, _dfmf.run_fits_concurrent(nat.SKF_DFMF, R, None, Theta, object_types, rank,
                                                       self.max_iter, self.dtype, G0, None, n_streams))
            return self
        local = {k: _dfmf.dfmf(G0=G0[k], **kw)
                 for k in my_runs(self.n_run)}          # one restart per GPU when distributed
        store_runs(self, gather_runs(local, self.n_run))
        return self


class DfmfTransform(FusionTransform):
    """Online transformer of new objects into a fitted fused space.

    Parameters (identical to the reference): max_iter=100, init_type=None (use the fuser's),
    n_run=1, stopping=None, stopping_system=None, fill_value=0, verbose=0, compute_err=False,
    callback=None, random_state=None, n_jobs=1.  Addition: dtype.
    """

    def __init__(self, max_iter=100, init_type=None, n_run=1, stopping=None,
                 stopping_system=None, fill_value=0, verbose=0, compute_err=False,
                 callback=None, random_state=None, n_jobs=1, dtype='f64'):
        super(DfmfTransform, self).__init__()
        self._set_params(vars())

    def transform(self, target, fusion_graph, fuser):
        self.target = target
        self.fusion_graph = fusion_graph
        self.fuser = fuser
        self._validate_graph()
        init_type = self.init_type if self.init_type is not None else fuser.init_type
        self.random_state = _random_state(self.random_state)
        rank = {ot: int(ot.rank) for ot in fusion_graph.object_types}

        # dfmf.py:176-189: preprocess, fill masked / non-finite entries with `fill_value`
        R, Theta = {}, {}
        for row_type, col_type in product(fusion_graph.object_types, repeat=2):
            for relation in fusion_graph.get_relations(row_type, col_type):
                data = relation.preprocessor(relation.data) if relation.preprocessor \
                    else relation.data
                if np.ma.is_masked(data):
                    data.fill_value = self.fill_value
                    data = data.filled()
                data[~np.isfinite(data)] = self.fill_value
                dest = R if relation.row_type != relation.col_type else Theta
                dest.setdefault((relation.row_type, relation.col_type), []).append(data)

        self.factors_ = defaultdict(list)
        if self.n_run > 1 and not (self.callback or self.stopping or self.stopping_system or self.compute_err):
            # the fold-ins into the models of all restarts share the uploads of the new relations and every launch
            # (reference: one joblib task per restart, dfmf.py:191-199)
            models = []
            for run in range(self.n_run):
                G = {(ot, ot): fuser.factor(ot, run) for ot in fuser.fusion_graph.object_types}
                S = {(rel.row_type, rel.col_type): [fuser.backbone(rel, run)]
                     for rel in fuser.fusion_graph.relations if rel.row_type != rel.col_type}
                models.append((G, S))
            self.factors_[target] = _dfmf.transform_runs(R, Theta, target, rank, models, max_iter=self.max_iter,
                                                         init_type=init_type, random_state=self.random_state,
                                                         dtype=self.dtype)
            return self
        for run in range(self.n_run):
            # frozen model of this run (dfmf.py:109-115); only the LAST relation of a type pair
            # survives in S there -- kept: one backbone per pair
            G = {(ot, ot): fuser.factor(ot, run) for ot in fuser.fusion_graph.object_types}
            S = {(rel.row_type, rel.col_type): [fuser.backbone(rel, run)]
                 for rel in fuser.fusion_graph.relations if rel.row_type != rel.col_type}
            G_new = _dfmf.transform(R_ij=R, Theta_i=Theta, target_obj_type=target,
                                    obj_type2rank=rank, G=G, S=S, max_iter=self.max_iter,
                                    init_type=init_type, stopping=self.stopping,
                                    stopping_system=self.stopping_system, verbose=self.verbose,
                                    compute_err=self.compute_err, callback=self.callback,
                                    random_state=self.random_state, dtype=self.dtype)
            self.factors_[target].append(G_new)
        return self
