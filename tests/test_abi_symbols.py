"""The C-ABI boundary on a machine without a GPU: libskfusion_hip.so (cross-compiled for gfx950 by
__graft_entry__.build()) loads, exports every function include/skfusion_hip.h declares, the ctypes
prototypes cover exactly that set, and the entry points that need no device answer sensibly.
No compute call is made here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import skfusion_amd._native as nat                     # noqa: E402

HEADER = os.path.join(ROOT, 'include', 'skfusion_hip.h')


def declared():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(skf_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(nat.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return nat.load_library()


def test_header_is_plain_c_and_cites_the_reference():
    subprocess.check_call(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', HEADER])
    text = open(HEADER).read()
    assert 'extern "C"' in text
    for cite in ('_dfmf.py:127', '_dfmc.py', 'dfmf.py'):        # reference file:line citations
        assert cite in text
    code = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    assert 'torch' not in code.lower() and 'tensor' not in code.lower()      # plain pointers and sizes only


def test_library_exports_every_declared_symbol(lib):
    names = declared()
    assert len(names) >= 26
    for name in names:
        assert hasattr(lib, name), '%s declared in the header but not exported' % name
    assert sorted(nat.SIGNATURES) == names               # the Python binding covers exactly the header


def test_exported_symbols_are_unmangled_c_linkage():
    out = subprocess.check_output(['nm', '-D', '--defined-only', nat.LIB_PATH]).decode()
    exported = set(line.split()[-1] for line in out.splitlines() if ' T ' in line)
    for name in declared():
        assert name in exported


def test_entry_points_without_a_device(lib):
    assert b'gfx950' in lib.skf_version()
    nbytes = C.c_size_t()
    assert lib.skf_pinv_sym_workspace_bytes(16, C.byref(nbytes)) == 0 and nbytes.value > 0
    # argument validation happens before any HIP call: status code + message, nothing thrown
    plan = nat._P()
    assert lib.skf_plan_create(0, None, 0, None, 0, None, None, C.byref(plan)) == -1     # SKF_E_INVALID
    assert b'null' in lib.skf_last_error() or b'object types' in lib.skf_last_error()
    off, n, dt = C.c_size_t(), C.c_size_t(), C.c_int32()
    assert lib.skf_exchange_range(None, 0, C.byref(off), C.byref(n), C.byref(dt)) == -1
    assert lib.skf_plan_destroy(None) in (0, -1)


def test_product_runtime_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    nat._runtime = None
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        nat.get_runtime()


def test_round5_entry_points_without_a_device(lib):
    """ABI version 5 adds entry points only: the limits of the small-graph schedule (what the host's `shared_launches` rule
    reads instead of its own copy of the constants), what a communicator is, and the launch counter."""
    assert lib.skf_abi_version() == nat.SKF_ABI_VERSION == 5
    mr, mo, mt, ml, mc, dv = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.skf_small_graph_limits(C.byref(mr), C.byref(mo), C.byref(mt), C.byref(ml), C.byref(mc), C.byref(dv)) == 0
    assert (mr.value, mo.value, dv.value) == (64, 8192, 16) and mt.value >= 3 and ml.value >= 3 and mc.value >= 1
    assert lib.skf_small_graph_limits(None, None, None, None, None, None) == 0            # null pointers are skipped
    comm = nat._P()
    assert lib.skf_comm_create(None, 0, 1, C.byref(comm)) == 0                            # one rank: no transport needed
    r, w, k, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.skf_comm_info(comm, C.byref(r), C.byref(w), C.byref(k), C.byref(n)) == 0
    assert (r.value, w.value, k.value, n.value) == (0, 1, nat.SKF_COMM_SINGLE, 1)
    assert lib.skf_comm_destroy(comm) == 0
    assert lib.skf_comm_create_null(3, 8, C.byref(comm)) == 0
    assert lib.skf_comm_info(comm, C.byref(r), C.byref(w), C.byref(k), C.byref(n)) == 0
    assert (r.value, w.value, k.value, n.value) == (3, 8, nat.SKF_COMM_NULL, 0)
    assert lib.skf_comm_destroy(comm) == 0
    assert lib.skf_comm_create(None, 0, 2, C.byref(comm)) == -1                           # two ranks need the unique id
    assert lib.skf_comm_info(None, None, None, None, None) == -1
    count = C.c_int64(-1)
    assert lib.skf_launch_count(C.byref(count)) == 0 and count.value >= 0
    assert lib.skf_split_clamps(C.byref(count)) == 0 and count.value == 0
