"""Emulated runtime for the CPU test-suite (TEST INFRASTRUCTURE, not a product path).

Compiles the UNMODIFIED kernel / C-ABI sources of ``scikit-fusion_amd/csrc`` with host
clang++ against ``tests/emul/include/hip/hip_runtime.h`` (a fiber-based SIMT emulator) into
``tests/emul/_build/libskf_emul.so`` and pairs it with plain host memory.  Tests install it
with ``use_runtime(...)`` (below) to check kernel index logic, tiling, the launch
schedule and the C ABI without a GPU.  The product runtime never loads this library.
"""
import contextlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC_DIR = os.path.join(ROOT, 'scikit-fusion_amd', 'csrc')
OUT = os.path.join(HERE, '_build', 'libskf_emul.so')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def _sources():
    deps = [os.path.join(SRC_DIR, f) for f in sorted(os.listdir(SRC_DIR))]
    deps.append(os.path.join(HERE, 'include', 'hip', 'hip_runtime.h'))
    deps.append(os.path.join(HERE, 'include', 'skf_asm.h'))
    deps.append(os.path.join(ROOT, 'include', 'skfusion_hip.h'))
    return deps


def build(force=False):
    if not os.path.exists(CLANG):
        raise RuntimeError('host clang++ not found at %s' % CLANG)
    newest = max(os.path.getmtime(p) for p in _sources())

    def fresh():
        return os.path.exists(OUT) and os.path.getmtime(OUT) >= newest
    if not force and fresh():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # several test processes (pytest-xdist workers) may get here at once: one builds, the others wait and find it done
    import fcntl
    with open(OUT + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not fresh():
            tmp = '%s.%d.tmp' % (OUT, os.getpid())
            cmd = [CLANG, '-x', 'c++', '-std=c++17', '-O2', '-fPIC', '-shared', '-Wno-psabi',
                   '-Wno-unused-variable', '-I', os.path.join(HERE, 'include'),
                   '-include', os.path.join(HERE, 'include', 'skf_asm.h'),      # host stand-ins ahead of csrc/skf_asm.h
                   os.path.join(SRC_DIR, 'skf_api.hip'), '-o', tmp]
            subprocess.check_call(cmd)
            os.replace(tmp, OUT)
    return OUT


class HostBuffer(object):
    __slots__ = ('ptr', 'nbytes', 'owner')

    def __init__(self, arr):
        self.owner = arr
        self.ptr = arr.ctypes.data
        self.nbytes = arr.nbytes


class HostMemory(object):
    """'Device' memory of the emulator = 256-byte aligned host arrays."""
    stream = None

    def empty(self, nbytes):
        raw = np.empty(int(nbytes) + 512, dtype=np.uint8)
        off = (-raw.ctypes.data) % 256
        view = raw[off:off + max(int(nbytes), 1)]
        view[:] = 0xA5                      # poison: catch reads of unwritten workspace
        return HostBuffer(view)

    def from_host(self, array, sync=True):
        a = np.ascontiguousarray(array)
        buf = self.empty(a.nbytes)
        buf.owner[:a.nbytes] = a.view(np.uint8).reshape(-1)
        return buf

    def to_host(self, buf, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return buf.owner[:n].copy().view(dtype).reshape(shape)

    def synchronize(self):
        pass

    def as_tensor(self, buf, offset, nbytes, np_dtype):
        import torch
        return torch.from_numpy(buf.owner[offset:offset + nbytes].view(np_dtype))


def emulated_runtime():
    import skfusion_amd._native as nat
    lib = nat.load_library(build())
    return nat.Runtime(lib, HostMemory(), 'emul')


@contextlib.contextmanager
def use_runtime(rt):
    """Install `rt` as the runtime of skfusion_amd._native for the duration of the block (test hook: the package itself
    has none -- its own runtime is the HIP library on a GPU or an error)."""
    import skfusion_amd._native as nat
    old, nat._runtime = nat._runtime, rt
    try:
        yield rt
    finally:
        nat._runtime = old
