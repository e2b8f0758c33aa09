"""ctypes binding of ``libskfusion_hip.so`` (C ABI: include/skfusion_hip.h) and the device-memory
plumbing (PyTorch-ROCm tensors used purely as HBM containers).

A *runtime* = (shared library, memory backend).  The product runtime is created lazily by
:func:`get_runtime` and FAILS LOUDLY when the HIP library or a GPU is missing -- there is no
CPU fallback in this package.  (The test-suite swaps the module-level runtime for an emulated one from
``tests/emul/runtime.py``; neither the emulator nor the hook that installs it is part of the package.)
"""
import ctypes as C
import os

import numpy as np

SKF_F64, SKF_F32, SKF_BF16 = 0, 1, 2
SKF_DFMF, SKF_DFMC, SKF_TRANSFORM = 0, 1, 2
SKF_ENGINE_MFMA, SKF_ENGINE_VALU = 0, 1
SKF_REL_ABSENT, SKF_REL_NO_COL_SIDE, SKF_REL_MASKED, SKF_REL_MASK_BITS, SKF_REL_BINARY = 1, 2, 4, 8, 16
SKF_REL_KNOWN_LISTS = 32
SKF_STAGE_CONTRACT, SKF_STAGE_BACKBONE, SKF_STAGE_ACCUMULATE, SKF_STAGE_UPDATE = 0, 1, 2, 3
SKF_X_W, SKF_X_Q, SKF_X_QM, SKF_X_ED = 0, 1, 2, 3
SKF_COMM_SINGLE, SKF_COMM_RCCL, SKF_COMM_CALLBACK, SKF_COMM_NULL = 0, 1, 2, 3
COMM_KIND = {0: 'single', 1: 'rccl', 2: 'callback', 3: 'null'}
SKF_OK, SKF_E_INVALID, SKF_E_STATE, SKF_E_WORKSPACE, SKF_E_HIP = 0, -1, -2, -3, -4

DTYPES = {'f64': SKF_F64, 'f32': SKF_F32, 'bf16': SKF_BF16, 'float64': SKF_F64, 'float32': SKF_F32}
NP_DTYPE = {SKF_F64: np.float64, SKF_F32: np.float32, SKF_BF16: np.float32}   # dtype of the masters


def to_bf16_bits(a):
    """float array -> uint16 bf16 bit patterns (round to nearest even), host side."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) >> 16).astype(np.uint16)


def from_bf16_bits(b):
    return (np.asarray(b, dtype=np.uint32) << 16).view(np.float32)

LIB_NAME = 'libskfusion_hip.so'
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', LIB_NAME)


class SkfNativeError(RuntimeError):
    """A call into libskfusion_hip.so returned an error status."""

    def __init__(self, code, message):
        super().__init__('libskfusion_hip: [%d] %s' % (code, message))
        self.code = code


class TypeDesc(C.Structure):
    _fields_ = [('n_obj', C.c_int64), ('rank', C.c_int32)]


class RelationDesc(C.Structure):
    _fields_ = [('row_type', C.c_int32), ('col_type', C.c_int32), ('data', C.c_void_p),
                ('ld', C.c_int64), ('mask', C.c_void_p), ('mask_ld', C.c_int64),
                ('row_begin', C.c_int64), ('n_rows', C.c_int64), ('flags', C.c_int32),
                ('known_bound', C.c_int64)]


class ThetaDesc(C.Structure):
    _fields_ = [('type', C.c_int32), ('data', C.c_void_p), ('ld', C.c_int64), ('nnz', C.c_int64)]


class Options(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('variant', C.c_int32), ('target_type', C.c_int32),
                ('engine', C.c_int32), ('part_index', C.c_int32), ('part_count', C.c_int32),
                ('flags', C.c_int32)]


SKF_OPT_OWNED_ROWS = 1
SKF_ABI_VERSION = 5          # include/skfusion_hip.h: the struct layouts above belong to this version


class GemmDesc(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('C2', C.c_void_p),
                ('mask', C.c_void_p),
                ('sa_m', C.c_int64), ('sa_k', C.c_int64), ('sb_k', C.c_int64), ('sb_n', C.c_int64),
                ('ldc', C.c_int64), ('ldc2', C.c_int64), ('ldmask', C.c_int64),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('aop', C.c_int32), ('epi', C.c_int32), ('nan_to_num', C.c_int32),
                ('splits', C.c_int32), ('a_dtype', C.c_int32), ('b_dtype', C.c_int32)]


# skf_collective_fn: int (*)(void* user, int32_t op, void* buf, size_t count, int32_t dtype, void* stream)
COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p)

# every symbol include/skfusion_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    'skf_plan_create': (C.c_int, [C.c_int32, C.POINTER(TypeDesc), C.c_int32, C.POINTER(RelationDesc),
                                  C.c_int32, C.POINTER(ThetaDesc), C.POINTER(Options), C.POINTER(_P)]),
    'skf_plan_destroy': (C.c_int, [_P]),
    'skf_plan_workspace_bytes': (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    'skf_plan_bind_workspace': (C.c_int, [_P, _P, C.c_size_t, _P]),
    'skf_set_factor': (C.c_int, [_P, C.c_int32, _P, C.c_int64, _P]),
    'skf_get_factor': (C.c_int, [_P, C.c_int32, _P, C.c_int64, _P]),
    'skf_set_backbone': (C.c_int, [_P, C.c_int32, _P, C.c_int64, _P]),
    'skf_get_backbone': (C.c_int, [_P, C.c_int32, _P, C.c_int64, _P]),
    'skf_iterate': (C.c_int, [_P, C.c_int32, _P]),
    'skf_iterate_batch': (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int32, _P]),
    'skf_plan_batchable': (C.c_int, [_P, C.POINTER(C.c_int32)]),
    'skf_small_graph_limits': (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'skf_plan_set_graph': (C.c_int, [_P, C.c_int32]),
    'skf_accumulate': (C.c_int, [_P, _P]),
    'skf_apply_update': (C.c_int, [_P, _P]),
    'skf_accumulator_range': (C.c_int, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    'skf_stage': (C.c_int, [_P, C.c_int32, _P]),
    'skf_exchange_range': (C.c_int, [_P, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_int32)]),
    'skf_comm_unique_id': (C.c_int, [_P]),
    'skf_comm_create': (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    'skf_comm_create_callback': (C.c_int, [C.c_int32, C.c_int32, _P, _P, C.POINTER(_P)]),
    'skf_comm_create_null': (C.c_int, [C.c_int32, C.c_int32, C.POINTER(_P)]),
    'skf_comm_info': (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'skf_launch_count': (C.c_int, [C.POINTER(C.c_int64)]),
    'skf_split_clamps': (C.c_int, [C.POINTER(C.c_int64)]),
    'skf_comm_destroy': (C.c_int, [_P]),
    'skf_owned_rows': (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64)]),
    'skf_abi_version': (C.c_int, []),
    'skf_plan_set_comm': (C.c_int, [_P, _P]),
    'skf_iterate_dist': (C.c_int, [_P, C.c_int32, _P]),
    'skf_exchange_bytes': (C.c_int, [_P, C.c_int32, C.POINTER(C.c_size_t)]),
    'skf_relation_sqerr': (C.c_int, [_P, C.c_int32, _P, _P]),
    'skf_get_contraction': (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P]),
    'skf_plan_set_profiling': (C.c_int, [_P, C.c_int32]),
    'skf_plan_get_profile': (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]),
    'skf_gemm': (C.c_int, [C.c_int32, C.c_int32, C.POINTER(GemmDesc), _P, C.c_size_t, _P]),
    'skf_gemm_bf16': (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, _P, C.c_size_t, _P]),
    'skf_gemm_bf16_tn': (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, _P, C.c_size_t, _P]),
    'skf_gemm_bits': (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, _P, C.c_size_t, _P]),
    'skf_to_bf16': (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _P]),
    'skf_pinv_sym_workspace_bytes': (C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    'skf_pinv_sym': (C.c_int, [C.c_int32, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P, C.c_size_t, _P]),
    'skf_fill_uniform': (C.c_int, [C.c_int32, _P, C.c_int64, C.c_int64, C.c_int64, C.c_uint64,
                                   C.c_double, C.c_double, _P]),
    'skf_fill_unknown_workspace_bytes': (C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    'skf_fill_unknown': (C.c_int, [C.c_int32, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int32, C.c_double,
                                   _P, C.c_size_t, _P]),
    'skf_cast': (C.c_int, [C.c_int32, _P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int64, C.c_int64, _P]),
    'skf_last_error': (C.c_char_p, []),
    'skf_version': (C.c_char_p, []),
}


def load_library(path=LIB_PATH):
    """dlopen the C-ABI library and attach the prototypes of every declared entry point.
    (SKF_LIB_PATH overrides the in-tree build: A/B runs of experimental builds.)"""
    path = os.environ.get('SKF_LIB_PATH', path)
    if not os.path.exists(path):
        raise ImportError(
            '%s not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  skfusion_amd has no CPU fallback.' % (LIB_NAME, path))
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME
    # libamdhip64.so.7, the same as /opt/rocm's).  If torch is imported FIRST the dynamic loader
    # satisfies our DT_NEEDED with torch's already-loaded copy, so device pointers and stream
    # handles are shared; loaded the other way round the process ends up with two HSA runtimes
    # and the second one sees "no ROCm-capable device".
    try:
        import torch                                     # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.skf_abi_version() != SKF_ABI_VERSION:     # descriptors grew between versions: never call across them
        raise ImportError('%s has ABI version %d, this binding was written for %d -- rebuild the library'
                          % (path, lib.skf_abi_version(), SKF_ABI_VERSION))
    return lib


class Buffer(object):
    """A device allocation: raw address + the object that keeps it alive."""
    __slots__ = ('ptr', 'nbytes', 'owner')

    def __init__(self, ptr, nbytes, owner):
        self.ptr, self.nbytes, self.owner = ptr, nbytes, owner


class TorchDeviceMemory(object):
    """HBM through PyTorch-ROCm (allocator + H2D/D2H copies only)."""

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('skfusion_amd needs an AMD GPU (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        self.torch = torch
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        # the engine works on its own stream (hipGraph capture is impossible on the legacy default
        # stream); every hand-over to / from torch goes through a device-wide synchronize()
        self._stream = torch.cuda.Stream(device=self.device)

    @property
    def stream(self):
        return self._stream.cuda_stream

    def empty(self, nbytes):
        t = self.torch.empty(max(int(nbytes), 256), dtype=self.torch.uint8, device=self.device)
        return Buffer(t.data_ptr(), int(nbytes), t)

    def from_host(self, array, sync=True):
        a = np.ascontiguousarray(array)
        t = self.torch.from_numpy(a.view(np.uint8).reshape(-1)).to(self.device)
        if sync:                                     # (sync=False: the caller synchronises once for several uploads)
            self.torch.cuda.synchronize(self.device)     # visible to the engine stream
        return Buffer(t.data_ptr(), a.nbytes, t)

    def to_host(self, buf, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        host = buf.owner[:n].cpu().numpy()
        return host.view(dtype).reshape(shape).copy()

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def new_stream(self):
        """A further stream for a plan that runs concurrently with others (raw handle, keep-alive object)."""
        s = self.torch.cuda.Stream(device=self.device)
        return s.cuda_stream, s

    def stream_scope(self):
        """Context in which torch (RCCL collectives) works on the engine's stream: a collective issued
        inside is ordered after the engine's kernels and before the ones launched next -- no host
        synchronisation on the data path."""
        return self.torch.cuda.stream(self._stream)

    def as_tensor(self, buf, offset, nbytes, np_dtype):
        """Zero-copy torch view of a byte range of a device buffer (for RCCL collectives)."""
        tdt = {np.dtype(np.float32): self.torch.float32, np.dtype(np.float64): self.torch.float64,
               np.dtype(np.uint8): self.torch.uint8}[np.dtype(np_dtype)]
        return buf.owner[offset:offset + nbytes].view(tdt)


class Runtime(object):
    def __init__(self, lib, mem, name):
        self.lib, self.mem, self.name = lib, mem, name

    def check(self, status):
        if status != 0:
            msg = self.lib.skf_last_error()
            raise SkfNativeError(status, msg.decode() if msg else '?')

    def call(self, fname, *args):
        self.check(getattr(self.lib, fname)(*args))


_runtime = None


def get_runtime():
    """The product runtime: libskfusion_hip.so + GPU memory.  Raises if either is missing."""
    global _runtime
    if _runtime is None:
        mem = TorchDeviceMemory()                  # imports torch + initialises its HIP runtime
        _runtime = Runtime(load_library(), mem, 'hip')
    return _runtime
