"""Host-side driver of one solver run: builds a plan in ``libskfusion_hip.so`` from the
``R / Theta / M`` dictionaries that arrive at the reference's functional seam
(``dfmf(**params)``, reference _dfmf.py:127; ``dfmc``, _dfmc.py:181; ``transform``,
_dfmf.py:330), keeps every matrix resident in HBM for the whole ``max_iter`` loop and copies
``G`` / ``S`` back once at the end (or per iteration when a callback is installed).
"""
import os
import ctypes as C

import numpy as np

from . import _native as nat


class DeviceMatrix(object):
    """A row-major matrix that already lives in HBM (engine dtype), e.g. synthetic benchmark
    data generated on the device with ``fill_uniform``."""

    def __init__(self, buf, shape, ld=None, binary=False):
        self.buf, self.shape, self.ld = buf, tuple(shape), ld if ld is not None else shape[1]
        self.binary = bool(binary)      # the caller's promise that every entry is 0 or 1 (SKF_REL_BINARY; checked at bind)
        self.nnz = 0                    # constraints: an upper bound on the non-zero entries (0 = unknown / dense)
        self.known = 0                  # byte masks in HBM: the number of KNOWN entries (mask == 0); 0 = not counted

    def rows(self, begin, count, itemsize):
        """View of `count` rows from `begin` on (no copy; the parent buffer stays referenced)."""
        return DeviceMatrix(_BufferView(self.buf, begin * self.ld * itemsize), (count, self.shape[1]), self.ld,
                            self.binary)


class PackedMask(object):
    """A DFMC mask in HBM as packed bits: bit (c & 7) of byte [r * ld + (c >> 3)], 1 = unknown entry
    (SKF_REL_MASK_BITS, include/skfusion_hip.h)."""

    def __init__(self, buf, shape, ld, known=0):
        self.buf, self.shape, self.ld = buf, tuple(shape), ld
        self.known = int(known)         # number of KNOWN entries (mask == 0); 0 = not counted


def pack_mask(mask, mem):
    """Boolean host mask -> PackedMask (1/8 of the bytes cross PCIe; the engine reads one bit per entry)."""
    m = np.asarray(mask, dtype=bool)
    if m.ndim != 2:
        raise ValueError('mask is not a matrix')
    bits = np.ascontiguousarray(np.packbits(m, axis=1, bitorder='little'))
    return PackedMask(mem.from_host(bits), m.shape, bits.shape[1], known=m.size - int(np.count_nonzero(m)))


def is_binary_matrix(arr, chunk_rows=4096):
    """True when every entry of the host matrix is 0 or 1 (SKF_REL_BINARY).  Row chunks with an early exit: no
    full-size boolean temporaries, and a real-valued relation is rejected after the first chunk."""
    a = np.asarray(arr)
    if a.ndim != 2 or a.size == 0:
        return False
    for r0 in range(0, a.shape[0], chunk_rows):
        blk = a[r0:r0 + chunk_rows]
        if not bool(((blk == 0) | (blk == 1)).all()):
            return False
    return True


def device_matrix_from_tensor(t):
    """Wrap a contiguous 2-D torch tensor that lives on the engine's device (no copy).  The engine reads it on ITS stream:
    the work that produces the tensor must be complete (torch.cuda.synchronize()) before a plan is created from it."""
    assert t.dim() == 2 and t.is_contiguous()
    return DeviceMatrix(nat.Buffer(t.data_ptr(), t.numel() * t.element_size(), t), tuple(t.shape))


class _BufferView(object):
    def __init__(self, parent, offset):
        self.parent, self.ptr = parent, parent.ptr + offset


def fill_uniform(shape, seed, dtype='f32', scale=1.0, shift=0.0, runtime=None):
    """Device matrix of counter-based uniforms (bit-identical to oracle hash_uniform_matrix)."""
    rt = runtime or nat.get_runtime()
    code = {'f64': nat.SKF_F64, 'f32': nat.SKF_F32, 'bf16': nat.SKF_BF16}[dtype]
    esz = {'f64': 8, 'f32': 4, 'bf16': 2}[dtype]
    buf = rt.mem.empty(shape[0] * shape[1] * esz)
    rt.call('skf_fill_uniform', code, buf.ptr, shape[0], shape[1], shape[1], int(seed),
            float(scale), float(shift), rt.mem.stream)
    return DeviceMatrix(buf, shape)


def owned_rows(dtype, n_obj, part_index, part_count, runtime=None):
    """(begin, count, chunk): the rows of a type of `n_obj` objects that part `part_index` of `part_count` owns under
    SKF_OPT_OWNED_ROWS, and the rows per part of the padded layout (skf_owned_rows: the library is the one place that
    decides the boundaries)."""
    rt = runtime or nat.get_runtime()
    code = nat.DTYPES[dtype] if isinstance(dtype, str) else dtype
    b, c, ch = C.c_int64(), C.c_int64(), C.c_int64()
    rt.call('skf_owned_rows', code, int(n_obj), int(part_index), int(part_count), C.byref(b), C.byref(c), C.byref(ch))
    return b.value, c.value, ch.value


_SMALL_LIMITS = {}


def small_graph_limits(runtime=None):
    """The library's limits for the three-launch schedule of small graphs / shared launches of restarts
    (skf_small_graph_limits), read once per runtime."""
    rt = runtime or nat.get_runtime()
    if id(rt) not in _SMALL_LIMITS:
        mr, mo, mt, ml, mc, dv = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        rt.call('skf_small_graph_limits', C.byref(mr), C.byref(mo), C.byref(mt), C.byref(ml), C.byref(mc), C.byref(dv))
        _SMALL_LIMITS[id(rt)] = dict(max_rank=mr.value, max_objects=mo.value, max_types=mt.value, max_relations=ml.value,
                                     max_constraints=mc.value, constraint_nnz_divisor=dv.value)
    return _SMALL_LIMITS[id(rt)]


def known_lists_pay(known, rows, cols, rank_row, dtype):
    """Whether a masked relation of rows x cols entries, `known` of them known, is worth keeping as lists of its known entries
    (the library's own rule for whole relations, skf_plan_create: share of known entries x rank of the row type <= 4;
    SKF_DFMC_SPARSE=0 never, =1 up to a quarter known) -- for plans with owned rows, where the caller decides for all ranks
    alike (SKF_REL_KNOWN_LISTS: the flag fixes the partial-sum convention of Q and W, so the callers take rank 0's answer,
    `_distributed.same_on_all_ranks` -- an environment that differs between ranks must not split them)."""
    import os
    mode = os.environ.get('SKF_DFMC_SPARSE')
    cells = float(rows) * float(cols)
    if mode == '0' or known <= 0 or cells <= 0 or rank_row > 1024 or known > 2000000000:      # (the library's own limits)
        return False
    share = known / cells
    return share <= 0.25 and (mode == '1' or share * rank_row <= 4.0)


def _sparse_bound(nnz, n):
    """skf_theta_desc.nnz for a constraint with `nnz` non-zeros: the count itself when the matrix is sparse enough for
    the CSR path (<= n*n / the library's divisor, 16), 0 (dense product, as the reference) otherwise or when unknown."""
    nnz = int(nnz or 0)
    return max(nnz, 1) if 0 < nnz <= (int(n) * int(n)) // small_graph_limits()['constraint_nnz_divisor'] else 0


_FILL = {'mean': 0, 'row_mean': 1, 'col_mean': 2, 'const': 3}


def fill_unknown_device(data, mask, strategy, value=0.0, dtype='f64', runtime=None):
    """Host matrix (NaN / inf / masked entries unknown) -> DeviceMatrix of the engine dtype with the unknown entries
    imputed on the device (reference Relation.filled(), fusion_graph.py:464-510).  SKF_BF16: filled in f32, then
    rounded to bf16 on the device."""
    rt = runtime or nat.get_runtime()
    code = nat.DTYPES[dtype]
    work = nat.SKF_F32 if code == nat.SKF_BF16 else code
    npd = nat.NP_DTYPE[work]
    arr = np.ascontiguousarray(data, dtype=npd)
    if arr.ndim != 2:
        raise ValueError('relation data is not a matrix')
    rows, cols = arr.shape
    buf = rt.mem.from_host(arr)
    mbuf, mld = None, 0
    if mask is not None:
        m = np.ascontiguousarray(np.asarray(mask, dtype=bool).astype(np.uint8))
        if m.shape != arr.shape:
            raise ValueError('mask shape mismatch')
        mbuf, mld = rt.mem.from_host(m), cols
    need = C.c_size_t()
    rt.call('skf_fill_unknown_workspace_bytes', rows, cols, C.byref(need))
    ws = rt.mem.empty(need.value)
    rt.call('skf_fill_unknown', work, buf.ptr, cols, rows, cols, mbuf.ptr if mbuf is not None else None, mld,
            _FILL[strategy], float(value), ws.ptr, need.value, rt.mem.stream)
    if code == nat.SKF_BF16:
        out = rt.mem.empty(rows * cols * 2)
        rt.call('skf_to_bf16', out.ptr, cols, work, buf.ptr, cols, rows, cols, 0, rt.mem.stream)
        rt.mem.synchronize()
        return DeviceMatrix(out, (rows, cols))
    rt.mem.synchronize()
    return DeviceMatrix(buf, (rows, cols))


def upload_matrix(data, dtype='f64', runtime=None):
    """Host matrix -> DeviceMatrix of the engine dtype (bf16: rounded on the host, as upload_graph does)."""
    rt = runtime or nat.get_runtime()
    code = nat.DTYPES[dtype]
    arr = np.ascontiguousarray(data, dtype=nat.NP_DTYPE[code])
    if arr.ndim != 2:
        raise ValueError('relation data is not a matrix')
    up = nat.to_bf16_bits(arr) if code == nat.SKF_BF16 else arr
    return DeviceMatrix(rt.mem.from_host(up), arr.shape)


def _all_reduce_sum(t):
    """Sum a tensor view over the process group (RCCL for GPU tensors; a gloo group -- CPU tests,
    single-GPU smoke runs -- stages GPU tensors through the host)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def _exchange(mem, tensors, reduce=None):
    """All-reduce(sum) the given workspace views between two stages of a sharded iteration.
    RCCL: issued on the engine's own stream (stream order replaces host synchronisation).  gloo
    (CPU tests, single-GPU smoke runs) or a caller-supplied `reduce`: bracketed by device syncs."""
    import os
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return
    if reduce is None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size() <= 1 and not os.environ.get('SKF_FORCE_COLLECTIVES'):
            return
        if tensors[0].is_cuda and dist.get_backend() != 'gloo' and hasattr(mem, 'stream_scope'):
            with mem.stream_scope():
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return
        reduce = _all_reduce_sum
    mem.synchronize()
    for t in tensors:
        reduce(t)
    mem.synchronize()


_RCCL = {}          # (rank, world, identity of the default process group) -> communicator handle (None: RCCL not bound on every rank)


def _group_token(dist):
    """What tells one default process group from the next one with the same (rank, world): the name torch gives every group
    it creates, else the identity of the group object."""
    try:
        pg = dist.distributed_c10d._get_default_group()
        return getattr(pg, 'group_name', None) or id(pg)
    except Exception:
        return None


def release_rccl_comms(runtime=None):
    """Destroy the RCCL communicators this process created (atexit; call it before destroy_process_group when the process
    goes on to create another group -- a later group with the same rank / world never reuses a handle of an earlier one
    either way: the cache is keyed on the group)."""
    for key, comm in list(_RCCL.items()):
        _RCCL.pop(key, None)
        if comm:
            try:
                (runtime or nat.get_runtime()).lib.skf_comm_destroy(comm)
            except Exception:
                pass


def comm_info(rt, comm):
    """dict(rank, world, transport: 'single' | 'rccl' | 'callback' | 'null', transport_ranks) of a communicator handle
    (skf_comm_info; transport_ranks of an RCCL communicator is what ncclCommCount reports)."""
    r, w, k, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    rt.call('skf_comm_info', comm, C.byref(r), C.byref(w), C.byref(k), C.byref(n))
    return {'rank': r.value, 'world': w.value, 'transport': nat.COMM_KIND[k.value], 'transport_ranks': n.value}


def launch_count(runtime=None):
    """Kernel launches this thread has issued through the library so far (skf_launch_count); callers take differences."""
    rt = runtime or nat.get_runtime()
    n = C.c_int64()
    rt.call('skf_launch_count', C.byref(n))
    return n.value


def split_clamps(runtime=None):
    """Split-K launches of this thread whose slice count the plan's scratch could not hold (skf_split_clamps): zero unless
    the sizing rule of skf_plan_create has fallen behind the launch-time pickers."""
    rt = runtime or nat.get_runtime()
    n = C.c_int64()
    rt.call('skf_split_clamps', C.byref(n))
    return n.value


def _shared_rccl_comm(rt, dist):
    """The process's RCCL communicator for the CURRENT default process group (created once per group, reused by every plan
    and restart), or None when it cannot be had on EVERY rank -- all ranks then agree on the callback path instead of some
    blocking in a broadcast the others never reach.  A failure is remembered for that group only.  Collective: every rank
    of the group must call it."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    key = (rank, world, _group_token(dist))
    if key in _RCCL:
        return _RCCL[key]
    # (communicators made for EARLIER process groups stay alive until release_rccl_comms() / process exit: a DevicePlan created
    #  under such a group may still hold the handle, and a destroyed handle handed to skf_iterate_dist is a use after free --
    #  a process that cycles through many groups keeps one idle communicator per group, which is the cheaper mistake)
    if not getattr(_shared_rccl_comm, '_atexit', False):
        import atexit
        atexit.register(release_rccl_comms)
        _shared_rccl_comm._atexit = True
    import logging
    log = logging.getLogger('skfusion_amd')
    where = 'cuda' if dist.get_backend() != 'gloo' else 'cpu'

    def all_agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=where)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    # 1. every rank checks LOCALLY that it can resolve RCCL (ncclGetUniqueId needs no peer) and the ranks agree BEFORE anyone
    #    enters ncclCommInitRank: that call is a rendezvous, and a rank that cannot load the library would leave the others
    #    waiting in it until the transport's own timeout
    mine, ident = None, [None]
    try:
        buf = (C.c_char * 128)()
        rt.call('skf_comm_unique_id', buf)
        mine = bytes(buf)
    except nat.SkfNativeError as exc:
        log.warning('RCCL not bound on rank %d (%s): collectives through torch.distributed', rank, exc)
    if not all_agree(mine is not None):
        _RCCL[key] = None
        return None
    if rank == 0:
        ident[0] = mine
    dist.broadcast_object_list(ident, src=0)
    # 2. the rendezvous itself, bounded: a rank still inside ncclCommInitRank after SKF_COMM_TIMEOUT seconds (default 60)
    #    gives up loudly instead of hanging until the NCCL watchdog fires
    import threading
    comm, box = nat._P(), {}

    device = torch.cuda.current_device() if torch.cuda.is_available() else None

    def create():
        try:
            if device is not None:
                torch.cuda.set_device(device)       # the current device is per THREAD: RCCL binds the communicator to it
            raw = (C.c_char * 128).from_buffer_copy(ident[0])
            rt.call('skf_comm_create', raw, rank, world, C.byref(comm))
            box['ok'] = True
        except nat.SkfNativeError as exc:
            box['error'] = exc
    limit = float(os.environ.get('SKF_COMM_TIMEOUT', '60'))
    th = threading.Thread(target=create, name='skf_comm_create', daemon=True)
    th.start()
    th.join(limit)
    if th.is_alive():
        # (the peers that did get through are now in the all_agree all-reduce below and would wait there for this rank: the
        #  caller must treat this error as fatal for the process group -- let it propagate and the launcher tears the job down)
        raise RuntimeError('skf_comm_create: rank %d of %d still waits in ncclCommInitRank after %.0f s -- the ranks disagree '
                           '(a peer failed or never called it); SKF_COMM_TIMEOUT sets the limit' % (rank, world, limit))
    ok = bool(box.get('ok'))
    if not ok:
        log.warning('skf_comm_create failed on rank %d (%s)', rank, box.get('error'))
    if not all_agree(ok):
        if ok:
            rt.lib.skf_comm_destroy(comm)
        _RCCL[key] = None
        return None
    _RCCL[key] = comm
    return comm


def _torch_collective(mem, ws, dist, rank, world):
    """skf_collective_fn over torch.distributed on views of the workspace `ws`: the transport of gloo groups (CPU tests,
    smoke runs: device views are staged through the host) and the fall-back of a group whose backend takes device tensors
    (nccl = RCCL) when the library could not bind RCCL itself on every rank, or `force_callback` -- there the collective
    runs on the device view directly (the nccl backend rejects host tensors)."""
    def collective(user, op, buf, count, dtype, stream):
        try:
            import torch
            if dtype == nat.SKF_BF16:           # bf16 rows travel as bytes (all-gather only; gloo has no 16-bit integer type)
                npd, count = np.uint8, count * 2
            else:
                npd = nat.NP_DTYPE[dtype]
            es = np.dtype(npd).itemsize
            n = count * (1 if op == 0 else world)
            view = mem.as_tensor(ws, int(buf) - ws.ptr, n * es, npd)
            mem.synchronize()
            on_device = view.is_cuda and dist.get_backend() != 'gloo'
            host = view if on_device else (view.cpu() if view.is_cuda else view)
            if op in (0, 1):              # (reduce-scatter: the all-reduce of the whole buffer covers the owned range)
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
            elif on_device:
                dist.all_gather_into_tensor(host, host[rank * count:(rank + 1) * count].clone())
            else:
                parts = [torch.empty(count, dtype=host.dtype) for _ in range(world)]
                dist.all_gather(parts, host[rank * count:(rank + 1) * count].clone())
                host = torch.cat(parts)
            if on_device:
                torch.cuda.synchronize()      # the backend's own stream: the engine's next launch must see the result
            elif view.is_cuda or host is not view:
                view.copy_(host)
            mem.synchronize()
            return 0
        except Exception:                 # never unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1
    return collective


class DevicePlan(object):
    """One (run, device) plan: relations + constraints uploaded, workspace bound."""

    def __init__(self, obj_types, n_obj, rank, relations, thetas, variant, dtype='f64',
                 target=None, engine=None, runtime=None, part=None, stream=None, sparse_known=None, owned=False):
        """relations: list of (row_type, col_type, ndarray, mask-or-None[, block]);
        thetas: list of (type, ndarray).  `block` (row-block sharding, `_distributed.partition_rows`)
        = dict(row_begin, n_rows, absent, col_side, masked): data / mask then hold only the local rows
        (None when absent); `part` = (index, count) of this plan among the row-block plans.
        `owned`: the row blocks are the ranges of `owned_rows` (SKF_OPT_OWNED_ROWS: every rank owns the same share of the rows
        of every type; such a plan iterates through iterate_dist only).
        `sparse_known` (DFMC): None = the engine decides from the number of known entries of every masked relation
        whether to keep only those (skf_relation_desc.known_bound); False = always the completed dense copy."""
        self.rt = runtime or nat.get_runtime()
        # `stream`: (raw handle, keep-alive) of a stream of its own for this plan (concurrent restarts);
        # default: the runtime's engine stream
        self._own_stream = stream
        self.stream = stream[0] if stream is not None else self.rt.mem.stream
        self.dtype = nat.DTYPES[dtype] if isinstance(dtype, str) else dtype
        if self.dtype not in nat.NP_DTYPE:
            raise ValueError('unsupported engine dtype %r' % (dtype,))
        self.np_dtype = nat.NP_DTYPE[self.dtype]
        self.types = list(obj_types)
        self.index = {t: k for k, t in enumerate(self.types)}
        self.n_obj = [int(n_obj[t]) for t in self.types]
        self.rank = [int(rank[t]) for t in self.types]
        self.relations = relations
        self.handle = nat._P()
        self._keep, self._keep_rel = [], []
        mem, lib = self.rt.mem, self.rt.lib

        tdesc = (nat.TypeDesc * len(self.types))()
        for k in range(len(self.types)):
            tdesc[k].n_obj, tdesc[k].rank = self.n_obj[k], self.rank[k]
        rdesc = (nat.RelationDesc * max(len(relations), 1))()
        for k, rel in enumerate(relations):
            i, j, data, mask = rel[:4]
            block = rel[4] if len(rel) > 4 else None
            rdesc[k].row_type, rdesc[k].col_type = self.index[i], self.index[j]
            rows_here = n_obj[i]
            if block is not None:
                rows_here = int(block['n_rows'])
                rdesc[k].row_begin, rdesc[k].n_rows = int(block['row_begin']), rows_here
                rdesc[k].flags = ((nat.SKF_REL_ABSENT if block.get('absent') else 0) |
                                  (0 if block.get('col_side', True) else nat.SKF_REL_NO_COL_SIDE) |
                                  (nat.SKF_REL_MASKED if block.get('masked') else 0))
                if block.get('absent'):
                    if block.get('known_lists'):
                        rdesc[k].flags |= nat.SKF_REL_KNOWN_LISTS
                    continue
            if isinstance(data, DeviceMatrix):
                arr, buf, ld = data, data.buf, data.ld
                if data.binary and mask is None and self.dtype == nat.SKF_BF16:
                    rdesc[k].flags |= nat.SKF_REL_BINARY
            else:
                arr = np.ascontiguousarray(data, dtype=self.np_dtype)
                if arr.ndim != 2:
                    raise ValueError('relation %d is not a matrix' % k)
                # SKF_BF16: a 0 / 1 relation is kept as a bitmap on the device (1/16 of the bytes per iteration).  A row
                # block carries the verdict on the WHOLE relation (`binary`), so that every rank of a row-sharded fit
                # takes the same contraction path for it
                if self.dtype == nat.SKF_BF16 and mask is None:
                    binary = block['binary'] if (block is not None and 'binary' in block) else is_binary_matrix(arr)
                    if binary:
                        rdesc[k].flags |= nat.SKF_REL_BINARY
                # SKF_BF16: relations are handed over as bf16 bit patterns
                up = nat.to_bf16_bits(arr) if self.dtype == nat.SKF_BF16 else arr
                buf, ld = mem.from_host(up), arr.shape[1]
            if tuple(arr.shape) != (rows_here, n_obj[j]):
                raise ValueError('relation (%s,%s) dimension mismatch: %r vs object counts (%d,%d)'
                                 % (i, j, tuple(arr.shape), rows_here, n_obj[j]))
            self._keep_rel.append(buf)
            rdesc[k].data, rdesc[k].ld = buf.ptr, ld
            if isinstance(mask, PackedMask):             # packed bits already in HBM (upload_graph)
                if tuple(mask.shape) != tuple(arr.shape):
                    raise ValueError('mask shape mismatch for relation (%s,%s)' % (i, j))
                self._keep_rel.append(mask.buf)
                rdesc[k].mask, rdesc[k].mask_ld = mask.buf.ptr, mask.ld
                rdesc[k].flags |= nat.SKF_REL_MASK_BITS
                rdesc[k].known_bound = mask.known
            elif isinstance(mask, DeviceMatrix):         # uint8 bytes already in HBM (device-generated data)
                if tuple(mask.shape) != tuple(arr.shape):
                    raise ValueError('mask shape mismatch for relation (%s,%s)' % (i, j))
                self._keep_rel.append(mask.buf)
                rdesc[k].mask, rdesc[k].mask_ld = mask.buf.ptr, mask.ld
                rdesc[k].known_bound = mask.known
            elif mask is not None:
                pm = pack_mask(mask, mem)
                if pm.shape != arr.shape:
                    raise ValueError('mask shape mismatch for relation (%s,%s)' % (i, j))
                self._keep_rel.append(pm.buf)
                rdesc[k].mask, rdesc[k].mask_ld = pm.buf.ptr, pm.ld
                rdesc[k].flags |= nat.SKF_REL_MASK_BITS
                rdesc[k].known_bound = pm.known
            if sparse_known is False or (block is not None and not block.get('known_lists')):
                rdesc[k].known_bound = 0
            if block is not None and block.get('known_lists'):       # (row ownership: decided for all ranks alike by the caller)
                rdesc[k].flags |= nat.SKF_REL_KNOWN_LISTS
        hdesc = (nat.ThetaDesc * max(len(thetas), 1))()
        for k, (t, data) in enumerate(thetas):
            if isinstance(data, DeviceMatrix):           # master dtype, already in HBM
                if tuple(data.shape) != (n_obj[t], n_obj[t]):
                    raise ValueError('constraint on %s dimension mismatch' % (t,))
                self._keep.append(data.buf)
                hdesc[k].type, hdesc[k].data, hdesc[k].ld = self.index[t], data.buf.ptr, data.ld
                hdesc[k].nnz = _sparse_bound(getattr(data, 'nnz', 0), n_obj[t])
                continue
            arr = np.ascontiguousarray(data, dtype=self.np_dtype)
            if arr.shape != (n_obj[t], n_obj[t]):
                raise ValueError('constraint on %s dimension mismatch' % (t,))
            buf = mem.from_host(arr)
            self._keep.append(buf)
            hdesc[k].type, hdesc[k].data, hdesc[k].ld = self.index[t], buf.ptr, arr.shape[1]
            # a sparse constraint (lambda I, a few must-link pairs per object, dicty's ppi) is kept as CSR on the device
            hdesc[k].nnz = _sparse_bound(int(np.count_nonzero(arr)), n_obj[t])
        opt = nat.Options(self.dtype, variant, self.index[target] if target is not None else -1,
                          nat.SKF_ENGINE_MFMA if engine is None else engine,
                          part[0] if part else 0, part[1] if part else 0,
                          nat.SKF_OPT_OWNED_ROWS if owned else 0)
        self.owned = bool(owned)
        self.rt.call('skf_plan_create', len(self.types), tdesc, len(relations), rdesc, len(thetas),
                     hdesc, C.byref(opt), C.byref(self.handle))
        nbytes = C.c_size_t()
        self.rt.call('skf_plan_workspace_bytes', self.handle, C.byref(nbytes))
        self.workspace_bytes = nbytes.value
        self.ws = mem.empty(nbytes.value)
        self.rt.call('skf_plan_bind_workspace', self.handle, self.ws.ptr, nbytes.value, self.stream)
        self._scalar = mem.empty(8)

    def release_relation_data(self):
        """SKF_BF16: the relations (and every engine's masks) were copied into the engine's own layout at
        bind time; the caller's buffers are not referenced afterwards and may be dropped.  The f32 / f64
        engines keep reading unmasked relations in place: no-op for them."""
        if self.dtype == nat.SKF_BF16:
            self._keep_rel = []

    # -- factors ---------------------------------------------------------------------------
    def set_factor(self, t, G):
        k = self.index[t]
        if isinstance(G, DeviceMatrix):
            if G.shape != (self.n_obj[k], self.rank[k]):
                raise ValueError('factor of %s has shape %r' % (t, G.shape))
            self.rt.call('skf_set_factor', self.handle, k, G.buf.ptr, G.ld, self.stream)
            self.rt.mem.synchronize()
            return
        arr = np.ascontiguousarray(G, dtype=self.np_dtype)
        if arr.shape != (self.n_obj[k], self.rank[k]):
            raise ValueError('factor of %s has shape %r, expected %r'
                             % (t, arr.shape, (self.n_obj[k], self.rank[k])))
        buf = self.rt.mem.from_host(arr)
        self.rt.call('skf_set_factor', self.handle, k, buf.ptr, arr.shape[1], self.stream)
        self.rt.mem.synchronize()

    def set_factors(self, G, sync=True):
        """Every factor of {type: array} (or {(type, type): array}) with ONE device synchronisation instead of two per
        factor -- on small graphs the synchronisations of set-up and read-back cost as much as the iterations.
        sync=False: the caller synchronises before the plan is iterated (several plans set up together)."""
        keep = []
        for key, arr in G.items():
            t = key[0] if isinstance(key, tuple) else key
            k = self.index[t]
            arr = np.ascontiguousarray(arr, dtype=self.np_dtype)
            if arr.shape != (self.n_obj[k], self.rank[k]):
                raise ValueError('factor of %s has shape %r, expected %r' % (t, arr.shape, (self.n_obj[k], self.rank[k])))
            keep.append((k, self.rt.mem.from_host(arr, sync=False), arr.shape[1]))
        self.rt.mem.synchronize()                       # the uploads (torch's stream) before the engine's copies
        for k, buf, ld in keep:
            self.rt.call('skf_set_factor', self.handle, k, buf.ptr, ld, self.stream)
        self._pending_uploads = keep                    # alive until the copies have run
        if sync:
            self.rt.mem.synchronize()
            self._pending_uploads = None

    def fetch_results(self, types, n_rel):
        """Issues the copies of every factor of `types` and the first `n_rel` backbones into device buffers; the caller
        synchronises once (for any number of plans) and calls read_results."""
        out = []
        for t in types:
            k = self.index[t]
            shape = (self.n_obj[k], self.rank[k])
            buf = self.rt.mem.empty(shape[0] * shape[1] * np.dtype(self.np_dtype).itemsize)
            self.rt.call('skf_get_factor', self.handle, k, buf.ptr, shape[1], self.stream)
            out.append((buf, shape))
        for rel in range(n_rel):
            shape = self._backbone_shape(rel)
            buf = self.rt.mem.empty(shape[0] * shape[1] * np.dtype(self.np_dtype).itemsize)
            self.rt.call('skf_get_backbone', self.handle, rel, buf.ptr, shape[1], self.stream)
            out.append((buf, shape))
        return out

    def read_results(self, fetched):
        return [self.rt.mem.to_host(buf, shape, self.np_dtype).astype(np.float64) for buf, shape in fetched]

    def get_factor(self, t):
        k = self.index[t]
        shape = (self.n_obj[k], self.rank[k])
        buf = self.rt.mem.empty(shape[0] * shape[1] * np.dtype(self.np_dtype).itemsize)
        self.rt.call('skf_get_factor', self.handle, k, buf.ptr, shape[1], self.stream)
        self.rt.mem.synchronize()
        return self.rt.mem.to_host(buf, shape, self.np_dtype).astype(np.float64)

    def _backbone_shape(self, rel):
        i, j = self.relations[rel][0], self.relations[rel][1]
        return (self.rank[self.index[i]], self.rank[self.index[j]])

    def set_backbone(self, rel, S):
        shape = self._backbone_shape(rel)
        arr = np.ascontiguousarray(S, dtype=self.np_dtype)
        if arr.shape != shape:
            raise ValueError('backbone %d has shape %r, expected %r' % (rel, arr.shape, shape))
        buf = self.rt.mem.from_host(arr)
        self.rt.call('skf_set_backbone', self.handle, rel, buf.ptr, shape[1], self.stream)
        self.rt.mem.synchronize()

    def get_backbone(self, rel):
        shape = self._backbone_shape(rel)
        buf = self.rt.mem.empty(shape[0] * shape[1] * np.dtype(self.np_dtype).itemsize)
        self.rt.call('skf_get_backbone', self.handle, rel, buf.ptr, shape[1], self.stream)
        self.rt.mem.synchronize()
        return self.rt.mem.to_host(buf, shape, self.np_dtype).astype(np.float64)

    def get_contraction(self, rel, which, rows=None):
        """P = R G_j (which=0) or Q = R^T G_i (which=1) as the last iteration left it (verification accessor); a masked
        relation kept as lists of its known entries answers which=2 with the row-side product P S^T instead of P."""
        i, j = self.relations[rel][0], self.relations[rel][1]
        if which == 0:
            shape = (rows if rows is not None else self.n_obj[self.index[i]], self.rank[self.index[j]])
        elif which == 2:
            shape = (rows if rows is not None else self.n_obj[self.index[i]], self.rank[self.index[i]])
        else:
            shape = (self.n_obj[self.index[j]], self.rank[self.index[i]])
        buf = self.rt.mem.empty(shape[0] * shape[1] * np.dtype(self.np_dtype).itemsize)
        self.rt.call('skf_get_contraction', self.handle, rel, which, buf.ptr, shape[1], self.stream)
        self.rt.mem.synchronize()
        return self.rt.mem.to_host(buf, shape, self.np_dtype)

    def set_graph(self, enable=True):
        """Replay one captured hipGraph per iteration (concurrent restarts: include/skfusion_hip.h)."""
        self.rt.call('skf_plan_set_graph', self.handle, 1 if enable else 0)

    # -- the loop --------------------------------------------------------------------------
    def iterate(self, n_iters=1):
        self.rt.call('skf_iterate', self.handle, int(n_iters), self.stream)

    def batchable(self):
        """True when this plan can be one of the plans of iterate_batch (the schedule for small graphs)."""
        yes = C.c_int32(0)
        self.rt.call('skf_plan_batchable', self.handle, C.byref(yes))
        return bool(yes.value)

    @staticmethod
    def iterate_batch(plans, n_iters=1):
        """`n_iters` iterations of several plans of ONE graph at once (skf_iterate_batch: independent restarts of a small
        graph, every launch serving all of them) on the stream of the first plan.  Returns False, with nothing launched,
        when the plans do not batch (not the small-graph schedule, different graphs / engines): iterate them one by
        one."""
        plans = list(plans)
        first = plans[0]
        arr = (nat._P * len(plans))(*[p.handle for p in plans])
        try:
            first.rt.call('skf_iterate_batch', arr, len(plans), int(n_iters), first.stream)
        except nat.SkfNativeError as exc:
            if exc.code == nat.SKF_E_STATE:
                return False
            raise
        return True

    def synchronize(self):
        self.rt.mem.synchronize()

    # -- collectives behind the C ABI -----------------------------------------------------
    def attach_comm(self, force_callback=False):
        """Give the plan a communicator over the ranks of the torch.distributed process group, so that
        `skf_iterate_dist` issues the exchanges of a sharded iteration itself: RCCL bound by the library (backend nccl:
        torch.distributed only carries the 128-byte unique id from rank 0 to the others; ONE communicator per process,
        shared by all plans and restarts), or -- gloo groups (CPU tests, two ranks sharing one GPU in smoke runs),
        `force_callback`, or an RCCL that cannot be bound on EVERY rank -- a callback that runs the collective through
        torch.distributed on a view of the workspace.  Returns True when a communicator is attached."""
        import os
        try:
            import torch.distributed as dist
        except ImportError:
            return False
        if not (dist.is_available() and dist.is_initialized()):
            return False
        rank, world = dist.get_rank(), dist.get_world_size()
        if world <= 1 and not os.environ.get('SKF_FORCE_COLLECTIVES'):
            return False
        on_gpu = self.rt.name == 'hip'
        comm = None
        if dist.get_backend() != 'gloo' and on_gpu and not force_callback:
            comm = _shared_rccl_comm(self.rt, dist)
        if comm is not None:
            self._comm, self._comm_shared = comm, True
        else:
            self._comm = nat._P()
            self._comm_shared = False
            self._comm_fn = nat.COLLECTIVE_FN(_torch_collective(self.rt.mem, self.ws, dist, rank, world))   # keep the trampoline alive
            self.rt.call('skf_comm_create_callback', rank, world, C.cast(self._comm_fn, C.c_void_p), None, C.byref(self._comm))
        self.rt.call('skf_plan_set_comm', self.handle, self._comm)
        return True

    def comm_info(self):
        """What the attached communicator is (`_engine.comm_info`), None without one."""
        return comm_info(self.rt, self._comm) if getattr(self, '_comm', None) else None

    def attach_single_comm(self):
        """A genuine communicator of ONE rank without a transport (skf_comm_create(NULL, 0, 1)): every collective returns at
        once and the iteration is the real one -- what a fit sharded by ownership runs on when the process has no group."""
        self._comm = nat._P()
        self._comm_shared = False
        self.rt.call('skf_comm_create', None, 0, 1, C.byref(self._comm))
        self.rt.call('skf_plan_set_comm', self.handle, self._comm)

    def attach_null_comm(self, rank, world):
        """A communicator whose collectives do nothing: times the compute of rank `rank` of `world` of a sharded fit on one
        GPU (bench.py --emulate-rank); results are meaningless: the factors are never updated.  NOT for fits -- a single
        process takes attach_single_comm."""
        self._comm = nat._P()
        self._comm_shared = False
        self.rt.call('skf_comm_create_null', int(rank), int(world), C.byref(self._comm))
        self.rt.call('skf_plan_set_comm', self.handle, self._comm)

    def attach_callback_comm(self, rank, world, collective):
        """A communicator whose collectives are `collective(op, view, count, rank, world)` on tensor views of the workspace
        (in-process groups of plans: tests, tests/helpers.ThreadGroup)."""
        mem, ws = self.rt.mem, self.ws

        def trampoline(user, op, buf, count, dtype, stream):
            try:
                if dtype == nat.SKF_BF16:       # bf16 rows travel as bytes (all-gather only)
                    npd, count = np.uint8, count * 2
                else:
                    npd = nat.NP_DTYPE[dtype]
                n = count * (1 if op == 0 else world)
                view = mem.as_tensor(ws, int(buf) - ws.ptr, n * np.dtype(npd).itemsize, npd)
                collective(op, view, count, rank, world)
                return 0
            except Exception:                 # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._comm = nat._P()
        self._comm_shared = False
        self._comm_fn = nat.COLLECTIVE_FN(trampoline)
        self.rt.call('skf_comm_create_callback', int(rank), int(world), C.cast(self._comm_fn, C.c_void_p), None, C.byref(self._comm))
        self.rt.call('skf_plan_set_comm', self.handle, self._comm)

    def exchange_bytes(self, world):
        """Bytes one rank sends per iteration of the distributed iteration on a ring of `world` ranks."""
        b = C.c_size_t()
        self.rt.call('skf_exchange_bytes', self.handle, int(world), C.byref(b))
        return b.value

    def iterate_dist(self, n_iters=1):
        """Iterations of a sharded fit (whole relations or row blocks) with the exchanges issued by the library
        (include/skfusion_hip.h, skf_iterate_dist): all-reduce of W / Q, reduce-scatter of E and D, update of the owned
        range of G, all-gather of G."""
        self.rt.call('skf_iterate_dist', self.handle, int(n_iters), self.stream)

    def iterate_sharded(self, n_iters=1):
        """Iterations of a relation-sharded run: this plan holds only this rank's relations;
        the E / D accumulators are summed over the ranks before the replicated G update -- by the library's own
        collectives when a communicator is attached (attach_comm), else by one all-reduce per iteration through
        torch.distributed."""
        if getattr(self, '_comm', None):
            return self.iterate_dist(n_iters)
        off, nbytes = C.c_size_t(), C.c_size_t()
        self.rt.call('skf_accumulator_range', self.handle, C.byref(off), C.byref(nbytes))
        acc = self.rt.mem.as_tensor(self.ws, off.value, nbytes.value, self.np_dtype)
        for _ in range(int(n_iters)):
            self.rt.call('skf_accumulate', self.handle, self.stream)
            _exchange(self.rt.mem, [acc])
            self.rt.call('skf_apply_update', self.handle, self.stream)

    def _exchange_views(self):
        """Zero-copy tensor views of the four exchange ranges (None when empty)."""
        views = []
        for which in (nat.SKF_X_W, nat.SKF_X_Q, nat.SKF_X_QM, nat.SKF_X_ED):
            off, nbytes, dt = C.c_size_t(), C.c_size_t(), C.c_int32()
            self.rt.call('skf_exchange_range', self.handle, which, C.byref(off), C.byref(nbytes), C.byref(dt))
            views.append(self.rt.mem.as_tensor(self.ws, off.value, nbytes.value, nat.NP_DTYPE[dt.value])
                         if nbytes.value else None)
        return views

    def stage(self, which):
        self.rt.call('skf_stage', self.handle, int(which), self.stream)

    def iterate_rows(self, n_iters=1, reduce=None):
        """Iterations of a row-block-sharded run (every rank lists all relations, each with its row
        block): four stages with an all-reduce(sum) of W and Q, of the masked relations' Q (DFMC),
        and of E / D between them -- include/skfusion_hip.h `skf_stage`.  `reduce(tensor)` defaults
        to torch.distributed.all_reduce (RCCL on GPUs, gloo in the CPU tests); with a communicator attached
        (attach_comm) and no `reduce` the library issues the exchanges itself (skf_iterate_dist)."""
        if reduce is None and getattr(self, '_comm', None):
            return self.iterate_dist(n_iters)
        xw, xq, xqm, xed = self._exchange_views()
        mem, call, h = self.rt.mem, self.rt.call, self.handle

        def exchange(*tensors):
            _exchange(mem, tensors, reduce)
        for _ in range(int(n_iters)):
            call('skf_stage', h, nat.SKF_STAGE_CONTRACT, self.stream)
            exchange(xw, xq)
            call('skf_stage', h, nat.SKF_STAGE_BACKBONE, self.stream)
            if xqm is not None:
                exchange(xqm)
            call('skf_stage', h, nat.SKF_STAGE_ACCUMULATE, self.stream)
            exchange(xed)
            call('skf_stage', h, nat.SKF_STAGE_UPDATE, self.stream)

    def relation_sqerr(self, rel):
        """sum (R - G_i S G_j^T)^2 for relation index `rel` (device reduction, one f64 D2H)."""
        self.rt.call('skf_relation_sqerr', self.handle, rel, self._scalar.ptr, self.stream)
        self.rt.mem.synchronize()
        return float(self.rt.mem.to_host(self._scalar, (1,), np.float64)[0])

    def set_profiling(self, enable=True):
        self.rt.call('skf_plan_set_profiling', self.handle, 1 if enable else 0)

    def get_profile(self):
        """(total ms, launches, executed flops, relation bytes read as stored) of the launches that walk a relation
        since the last call."""
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        self.rt.call('skf_plan_get_profile', self.handle, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
        return ms.value, n.value, fl.value, by.value

    def close(self):
        if self.handle:
            self.rt.lib.skf_plan_destroy(self.handle)
            self.handle = nat._P()
        if getattr(self, '_comm', None):
            if not getattr(self, '_comm_shared', False):
                self.rt.lib.skf_comm_destroy(self._comm)
            self._comm = None
        self._keep, self._keep_rel = [], []
        self.ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upload_graph(rel_list, theta_list, dtype, runtime=None):
    """Relations / masks / constraints to HBM ONCE, as DeviceMatrix entries that any number of plans can
    share (concurrent restarts of one graph): same conversions as DevicePlan applies to host arrays."""
    rt = runtime or nat.get_runtime()
    code = nat.DTYPES[dtype] if isinstance(dtype, str) else dtype
    npd = nat.NP_DTYPE[code]
    rels, thetas = [], []
    for rel in rel_list:
        i, j, data, mask = rel[:4]
        if not isinstance(data, DeviceMatrix):
            arr = np.ascontiguousarray(data, dtype=npd)
            up = nat.to_bf16_bits(arr) if code == nat.SKF_BF16 else arr
            binary = code == nat.SKF_BF16 and mask is None and is_binary_matrix(arr)
            data = DeviceMatrix(rt.mem.from_host(up), arr.shape, binary=binary)
        if mask is not None and not isinstance(mask, (DeviceMatrix, PackedMask)):
            mask = pack_mask(mask, rt.mem)
        rels.append((i, j, data, mask) + tuple(rel[4:]))
    for t, data in theta_list:
        if not isinstance(data, DeviceMatrix):
            arr = np.ascontiguousarray(data, dtype=npd)
            data = DeviceMatrix(rt.mem.from_host(arr), arr.shape)
            data.nnz = int(np.count_nonzero(arr))          # (as DevicePlan counts a host array: the same path either way)
        thetas.append((t, data))
    return rels, thetas


def iterate_rows_lockstep(plans, n_iters=1):
    """Drive the row-block plans of ALL ranks from one process (same device), summing the exchange
    ranges by hand where the ranks of a process group would all-reduce: the single-GPU / emulator
    test vehicle of the row-block schedule."""
    views = [p._exchange_views() for p in plans]

    def exchange(*which):
        for p in plans:
            p.synchronize()
        for w in which:
            ts = [v[w] for v in views]
            if ts[0] is None:
                continue
            total = ts[0].clone()
            for t in ts[1:]:
                total += t
            for t in ts:
                t.copy_(total)
        for p in plans:
            p.synchronize()
    for _ in range(int(n_iters)):
        for p in plans:
            p.stage(nat.SKF_STAGE_CONTRACT)
        exchange(0, 1)
        for p in plans:
            p.stage(nat.SKF_STAGE_BACKBONE)
        exchange(2)
        for p in plans:
            p.stage(nat.SKF_STAGE_ACCUMULATE)
        exchange(3)
        for p in plans:
            p.stage(nat.SKF_STAGE_UPDATE)


def flatten_relations(R, M=None):
    """dict {(i,j): [matrix, ...]} -> [(i, j, matrix, mask)] in dict order, list order
    (the order the reference walks `R`, _dfmf.py:249-251)."""
    out = []
    for (i, j), mats in R.items():
        for l, m in enumerate(mats):
            mask = None
            if M is not None and M.get((i, j)) is not None:
                mask = M[i, j][l]
            out.append((i, j, m, mask))
    return out


def flatten_thetas(Theta):
    out = []
    for (i, i2), mats in Theta.items():
        for m in mats:
            out.append((i, m))
    return out


def count_objects(obj_types, R):
    """Objects per type from the relation shapes (reference count_objects, _dfmf.py:95-124).
    A mismatch is a hard error here (the reference only logs it and carries on)."""
    from .fusion.base import DataFusionError
    n = {}
    for (i, j), mats in R.items():
        for m in mats:
            for ax, t in enumerate((i, j)):
                have = n.setdefault(t, m.shape[ax])
                if have != m.shape[ax]:
                    raise DataFusionError('Relation matrix R_(%s,%s) dimension mismatch' % (i, j))
    if set(obj_types) != set(n):
        raise DataFusionError('Object type specification mismatch')
    return n


class DeviceReconstructor(object):
    """R_hat blocks = G_row[block] @ S @ G_col.T on the device (two strided MFMA GEMMs through ``skf_gemm``) with
    ``S`` and the column factor uploaded ONCE and kept resident across blocks -- the building block of
    ``FusionFit.complete_blocks`` for relations whose dense reconstruction does not fit on the host in one
    piece (reference base.py:119-146 materialises the whole product) and of the chained latent profiles
    (f4: ``S`` is then the product of the backbones along a ``chain()`` path).  ``G_col=None``: blocks are
    ``G_row[block] @ S`` (the profile form of reference examples/pharma_chaining.py:43-53, one GEMM)."""

    def __init__(self, S, G_col=None, dtype='f64', runtime=None, G_col_device=None):
        """``G_col_device``: the buffer another reconstructor of the same dtype already uploaded for this column factor
        (``other.b`` with ``other.nj`` rows: several paths that end in one object type share one copy in HBM)."""
        self.rt = runtime or nat.get_runtime()
        code = nat.DTYPES[dtype]
        self.code = nat.SKF_F32 if code == nat.SKF_BF16 else code
        self.npd = nat.NP_DTYPE[self.code]
        self.es = np.dtype(self.npd).itemsize
        Sm = np.ascontiguousarray(S, dtype=self.npd)
        B = None if G_col is None else np.ascontiguousarray(G_col, dtype=self.npd)
        if Sm.ndim != 2 or (B is not None and (B.ndim != 2 or B.shape[1] != Sm.shape[1])):
            raise ValueError('shape mismatch in reconstruction')
        self.ci, self.cj = Sm.shape
        self.nj = self.cj if B is None else B.shape[0]          # columns of a block
        self.s = self.rt.mem.from_host(Sm)
        shared = B is not None and G_col_device is not None
        self.b = None if B is None else (G_col_device if shared else self.rt.mem.from_host(B))
        self.uploads = 1 if (B is None or shared) else 2   # H2D copies of S / G_col so far (does not grow with the block count)
        self._h = self._out = None
        self._rows = 0

    def _gemm(self, Ap, sa_m, sa_k, Bp, sb_k, sb_n, Cp, ldc, M, N, K):
        _device_gemm(self.rt, self.code, Ap, sa_m, sa_k, Bp, sb_k, sb_n, Cp, ldc, M, N, K)

    def block(self, G_row_block, device=False):
        """Reconstruction of the rows of this block: a host ndarray (float64), or -- device=True -- a
        DeviceMatrix of the engine dtype that is valid until the next call (no D2H copy at all)."""
        A = np.ascontiguousarray(G_row_block, dtype=self.npd)
        if A.ndim != 2 or A.shape[1] != self.ci:
            raise ValueError('shape mismatch in reconstruction')
        m = A.shape[0]
        mem = self.rt.mem
        if m > self._rows:                     # scratch grows to the largest block seen, then is reused
            self._h = mem.empty(m * self.cj * self.es)
            self._out = self._h if self.b is None else mem.empty(m * self.nj * self.es)
            self._rows = m
        a = mem.from_host(A)
        self._gemm(a.ptr, self.ci, 1, self.s.ptr, self.cj, 1, self._h.ptr, self.cj, m, self.cj, self.ci)      # H = G_blk S
        if self.b is not None:
            self._gemm(self._h.ptr, self.cj, 1, self.b.ptr, 1, self.cj, self._out.ptr, self.nj, m, self.nj, self.cj)   # H G_col^T
        mem.synchronize()
        if device:
            return DeviceMatrix(self._out, (m, self.nj))
        return mem.to_host(self._out, (m, self.nj), self.npd).astype(np.float64)


def _device_gemm(rt, code, Ap, sa_m, sa_k, Bp, sb_k, sb_n, Cp, ldc, M, N, K):
    """C[M x N] = A[M x K] B[K x N] through skf_gemm (strides in elements; matrix-core GEMM of the master type `code`)."""
    d = nat.GemmDesc()
    d.A, d.B, d.C = Ap, Bp, Cp
    d.sa_m, d.sa_k, d.sb_k, d.sb_n, d.ldc, d.ldc2 = sa_m, sa_k, sb_k, sb_n, ldc, ldc
    d.M, d.N, d.K = M, N, K
    d.splits, d.a_dtype, d.b_dtype = 1, -1, -1
    rt.call('skf_gemm', code, nat.SKF_ENGINE_MFMA, C.byref(d), None, 0, rt.mem.stream)


def chain_backbone(backbones, runtime=None):
    """Product of the backbones along one `chain()` path, S_01 S_12 ... (c_first x c_last), on the device in f64 whatever
    the engine dtype (c x c work; the profile built from it inherits every digit lost here) -- left to right, the order of
    `reduce(np.dot, cf)` in reference examples/dicty_chaining.py:43-45 / pharma_chaining.py:46-47.  Returns a float64
    ndarray; an empty path has no backbone (the caller uses the factor itself)."""
    rt = runtime or nat.get_runtime()
    mats = [np.ascontiguousarray(S, dtype=np.float64) for S in backbones]
    if not mats:
        raise ValueError('an empty path has no backbone')
    for a, b in zip(mats[:-1], mats[1:]):
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[0]:
            raise ValueError('backbones along the path do not chain: %r then %r' % (a.shape, b.shape))
    if len(mats) == 1:
        return mats[0].copy()
    mem = rt.mem
    ups = [mem.from_host(S, sync=False) for S in mats]          # every buffer lives until the last product is through:
    mem.synchronize()                                           # the launches run on the engine's stream, not torch's
    keep, acc, rows, inner = [], ups[0], mats[0].shape[0], mats[0].shape[1]
    for S, nxt in zip(mats[1:], ups[1:]):
        out = mem.empty(rows * S.shape[1] * 8)
        keep.append(out)
        _device_gemm(rt, nat.SKF_F64, acc.ptr, inner, 1, nxt.ptr, S.shape[1], 1, out.ptr, S.shape[1], rows, S.shape[1], inner)
        acc, inner = out, S.shape[1]
    mem.synchronize()
    return mem.to_host(acc, (rows, inner), np.float64)


def device_reconstruct(G_row_block, S, G_col, dtype='f64', runtime=None):
    """One block (S and G_col uploaded for this call only; `DeviceReconstructor` keeps them resident)."""
    return DeviceReconstructor(S, G_col, dtype, runtime).block(G_row_block)
