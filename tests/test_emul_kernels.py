"""Kernel logic on the CPU: the real kernel sources run under the host SIMT emulator
(tests/emul) and are compared with NumPy.  These tests validate indexing / tiling / epilogues
/ the launch schedule; the `-m gpu` tests repeat them on the hardware."""
import ctypes as C

import numpy as np
import pytest

import skfusion_amd._native as nat
from emul.runtime import emulated_runtime
from helpers import relerr


@pytest.fixture(scope='module')
def rt():
    return emulated_runtime()


def run_gemm(rt, dtype, engine, A, B, transA=False, transB=False, aop=0, epi=0, C0=None, C20=None,
             mask=None, nan=0, splits=0, a_dtype=-1, b_dtype=-1):
    """C = epi(aop(op(A)) @ op(B)) through skf_gemm; A/B given as stored (row-major)."""
    npd = nat.NP_DTYPE[dtype]
    A = np.ascontiguousarray(A, dtype=npd if a_dtype < 0 else nat.NP_DTYPE[a_dtype])
    B = np.ascontiguousarray(B, dtype=npd if b_dtype < 0 else nat.NP_DTYPE[b_dtype])
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    N = B.shape[0] if transB else B.shape[1]
    mem = rt.mem
    a, b = mem.from_host(A), mem.from_host(B)
    Cm = np.zeros((M, N), npd) if C0 is None else np.array(C0, dtype=npd)
    C2m = np.zeros((M, N), npd) if C20 is None else np.array(C20, dtype=npd)
    c, c2 = mem.from_host(Cm), mem.from_host(C2m)
    d = nat.GemmDesc()
    d.A, d.B, d.C, d.C2 = a.ptr, b.ptr, c.ptr, c2.ptr
    d.sa_m, d.sa_k = (1, A.shape[1]) if transA else (A.shape[1], 1)
    d.sb_k, d.sb_n = (1, B.shape[1]) if transB else (B.shape[1], 1)
    d.ldc = d.ldc2 = N
    d.M, d.N, d.K = M, N, K
    d.aop, d.epi, d.nan_to_num, d.splits = aop, epi, nan, splits
    d.a_dtype, d.b_dtype = a_dtype, b_dtype
    keep = None
    if mask is not None:                    # the stand-alone operator takes one byte per entry
        keep = mem.from_host(np.ascontiguousarray(mask, dtype=np.uint8))
        d.mask, d.ldmask = keep.ptr, N
    ws = mem.empty(64 * M * N * 8 + 256)
    rt.call('skf_gemm', dtype, engine, C.byref(d), ws.ptr, ws.nbytes, None)
    return mem.to_host(c, (M, N), npd), mem.to_host(c2, (M, N), npd)


SHAPES = [(1, 1, 1), (5, 3, 7), (50, 10, 100), (64, 64, 32), (70, 130, 33), (129, 200, 65),
          (200, 129, 17), (33, 260, 40)]


@pytest.mark.parametrize('dtype', [nat.SKF_F64, nat.SKF_F32])
@pytest.mark.parametrize('engine', [nat.SKF_ENGINE_MFMA, nat.SKF_ENGINE_VALU])
@pytest.mark.parametrize('shape', SHAPES)
def test_gemm_plain_all_layouts(rt, dtype, engine, shape):
    M, N, K = shape
    rs = np.random.RandomState(M * 1000 + N * 10 + K)
    tol = 1e-13 if dtype == nat.SKF_F64 else 2e-6
    for transA in (False, True):
        for transB in (False, True):
            A = rs.randn(*((K, M) if transA else (M, K)))
            B = rs.randn(*((N, K) if transB else (K, N)))       # asymmetric operands
            got, _ = run_gemm(rt, dtype, engine, A, B, transA, transB)
            want = (A.T if transA else A) @ (B.T if transB else B)
            assert relerr(got, want) < tol, (transA, transB)


@pytest.mark.parametrize('dtype', [nat.SKF_F64, nat.SKF_F32])
@pytest.mark.parametrize('engine', [nat.SKF_ENGINE_MFMA, nat.SKF_ENGINE_VALU])
def test_gemm_epilogues_and_operand_ops(rt, dtype, engine):
    rs = np.random.RandomState(5)
    M, N, K = 70, 45, 90
    tol = 1e-13 if dtype == nat.SKF_F64 else 2e-6
    A, B = rs.randn(M, K), rs.randn(K, N)
    C0, C20 = rs.rand(M, N), rs.rand(M, N)
    P = A @ B
    got, _ = run_gemm(rt, dtype, engine, A, B, epi=1, C0=C0)
    assert relerr(got, C0 + P) < tol
    g1, g2 = run_gemm(rt, dtype, engine, A, B, epi=2, C0=C0, C20=C20)
    assert relerr(g1, np.maximum(P, 0)) < tol and relerr(g2, np.maximum(-P, 0)) < tol
    g1, g2 = run_gemm(rt, dtype, engine, A, B, epi=3, C0=C0, C20=C20)
    assert relerr(g1, C0 + np.maximum(P, 0)) < tol and relerr(g2, C20 + np.maximum(-P, 0)) < tol
    mask = rs.rand(M, N) > 0.6
    got, _ = run_gemm(rt, dtype, engine, A, B, epi=4, C0=C0, mask=mask)
    assert relerr(got, np.where(mask, P, C0)) < tol
    got, _ = run_gemm(rt, dtype, engine, A, B, aop=1)
    assert relerr(got, np.maximum(A, 0) @ B) < tol
    got, _ = run_gemm(rt, dtype, engine, A, B, aop=2)
    assert relerr(got, np.maximum(-A, 0) @ B) < tol


@pytest.mark.parametrize('dtype', [nat.SKF_F64, nat.SKF_F32])
def test_gemm_split_k_and_nan_to_num(rt, dtype):
    rs = np.random.RandomState(6)
    tol = 1e-13 if dtype == nat.SKF_F64 else 3e-6
    A, B = rs.randn(700, 20), rs.randn(700, 30)          # Gram-like: K = 700, tiny output
    want = A.T @ B
    for splits in (0, 1, 3, 7):
        got, _ = run_gemm(rt, dtype, nat.SKF_ENGINE_MFMA, A, B, transA=True, splits=splits)
        assert relerr(got, want) < tol, splits
    g1, g2 = run_gemm(rt, dtype, nat.SKF_ENGINE_MFMA, A, B, transA=True, splits=4, epi=3,
                      C0=np.ones((20, 30)), C20=np.ones((20, 30)))
    assert relerr(g1, 1 + np.maximum(want, 0)) < tol and relerr(g2, 1 + np.maximum(-want, 0)) < tol
    A2 = A.copy()
    A2[3, 2] = np.nan
    A2[5, 4] = np.inf
    got, _ = run_gemm(rt, dtype, nat.SKF_ENGINE_MFMA, A2, B, transA=True, nan=1, splits=2)
    with np.errstate(invalid='ignore'):
        ref = np.nan_to_num((A2.T @ B).astype(nat.NP_DTYPE[dtype]))
    assert np.isfinite(got).all()
    np.testing.assert_array_equal(got[2] == 0, ref[2] == 0)
    assert relerr(np.delete(got, [2, 4], axis=0), np.delete(ref, [2, 4], axis=0)) < tol


@pytest.mark.parametrize('engine', [nat.SKF_ENGINE_MFMA, nat.SKF_ENGINE_VALU])
def test_gemm_mixed_operand_types(rt, engine):
    """f32 operands with f64 arithmetic (Gram / G^T P) and f32 x f64 -> f32 (P S^T, G B)."""
    rs = np.random.RandomState(8)
    G = rs.rand(900, 40).astype(np.float32)
    P = rs.randn(900, 70).astype(np.float32)
    got, _ = run_gemm(rt, nat.SKF_F64, engine, G, P, transA=True, a_dtype=nat.SKF_F32, b_dtype=nat.SKF_F32)
    want = G.astype(np.float64).T @ P.astype(np.float64)
    assert got.dtype == np.float64 and relerr(got, want) < 1e-14
    S = rs.randn(40, 70)
    got, _ = run_gemm(rt, nat.SKF_F32, engine, P, S, transB=True, a_dtype=nat.SKF_F32, b_dtype=nat.SKF_F64)
    assert got.dtype == np.float32 and relerr(got, P.astype(np.float64) @ S.T) < 2e-6
    lib = rt.lib
    d = nat.GemmDesc()
    d.A = d.B = d.C = 1
    d.a_dtype, d.b_dtype = nat.SKF_F64, nat.SKF_F32
    d.M = d.N = d.K = 4
    assert lib.skf_gemm(nat.SKF_F32, engine, C.byref(d), None, 0, None) == -1


def run_gemm_bf16(rt, A, B, splits=0):
    """C = A @ B with bf16 operands through skf_gemm_bf16 (A: M x K, B: K x N given as floats)."""
    M, K = A.shape
    N = B.shape[1]
    Kp = (K + 63) // 64 * 64
    Ab = np.zeros((M, Kp), np.uint16)
    Ab[:, :K] = nat.to_bf16_bits(A)
    Bb = np.zeros((N, Kp), np.uint16)
    Bb[:, :K] = nat.to_bf16_bits(B.T)
    a, b = rt.mem.from_host(Ab), rt.mem.from_host(Bb)
    c = rt.mem.empty(M * N * 4)
    ws = rt.mem.empty(40 * M * N * 4 + 256)
    rt.call('skf_gemm_bf16', a.ptr, Kp, b.ptr, Kp, c.ptr, N, M, N, Kp, splits, ws.ptr, ws.nbytes, None)
    got = rt.mem.to_host(c, (M, N), np.float32)
    want = nat.from_bf16_bits(Ab[:, :K]).astype(np.float64) @ nat.from_bf16_bits(Bb[:, :K]).astype(np.float64).T
    return got, want


@pytest.mark.parametrize('shape', [(1, 1, 1), (16, 16, 32), (128, 128, 64), (130, 100, 70), (257, 128, 200),
                                   (100, 256, 129), (200, 200, 500), (129, 300, 64),
                                   (4100, 128, 70), (4200, 256, 200), (4097, 300, 64), (4100, 48, 200)])
def test_gemm_bf16_contraction(rt, shape):
    """Both workgroup shapes: the 128-row register-staged kernel (M < 4096) and the 256-row LDS-DMA
    kernel with its 3-deep ring and scheduled DMA pieces (M >= 4096); row / column / K tails, split-K."""
    M, N, K = shape
    rs = np.random.RandomState(M + N + K)
    A, B = rs.randn(M, K), rs.randn(K, N)           # asymmetric operands
    for splits in ((0, 1, 3) if M < 4096 else (0, 2)):
        got, want = run_gemm_bf16(rt, A, B, splits)
        assert relerr(got, want) < 2e-6, splits     # exact products, f32 accumulation


def run_gemm_bf16_tn(rt, R, G, splits=0):
    """Q = R^T @ G through skf_gemm_bf16_tn: R (K x M, row-major, the relation as stored) is read
    transposed out of LDS; G (K x N) is passed as the stored transpose G^T."""
    K, M = R.shape
    N = G.shape[1]
    Kp = (K + 63) // 64 * 64
    lda = (M + 63) // 64 * 64
    Rb = np.zeros((Kp, lda), np.uint16)
    Rb[:K, :M] = nat.to_bf16_bits(R)
    Gt = np.zeros((N, Kp), np.uint16)
    Gt[:, :K] = nat.to_bf16_bits(G.T)
    a, b = rt.mem.from_host(Rb), rt.mem.from_host(Gt)
    c = rt.mem.empty(M * N * 4)
    ws = rt.mem.empty(40 * M * N * 4 + 256)
    rt.call('skf_gemm_bf16_tn', a.ptr, lda, b.ptr, Kp, c.ptr, N, M, N, Kp, splits, ws.ptr, ws.nbytes, None)
    got = rt.mem.to_host(c, (M, N), np.float32)
    want = nat.from_bf16_bits(Rb[:K, :M]).astype(np.float64).T @ nat.from_bf16_bits(Gt[:, :K]).astype(np.float64).T
    return got, want


@pytest.mark.parametrize('shape', [(1, 1, 1), (16, 16, 32), (64, 128, 64), (100, 130, 70), (200, 257, 128),
                                   (129, 100, 256), (500, 200, 200), (64, 129, 300), (300, 520, 100), (200, 300, 40)])
def test_gemm_bf16_transposed_a(rt, shape):
    """Q = R^T G_i from the row-major relation: LDS-DMA of [64 k][256 m] tiles, ds_read_b64_tr_b16 fragments
    (emulated with the lane map measured on the hardware); K x M x N with tails everywhere, split-K."""
    K, M, N = shape
    rs = np.random.RandomState(M + N + K)
    R, G = rs.randn(K, M), rs.randn(K, N)           # asymmetric: a transposed or mis-swizzled read cannot pass
    for splits in (0, 1, 2):
        got, want = run_gemm_bf16_tn(rt, R, G, splits)
        assert relerr(got, want) < 2e-6, splits


def run_gemm_bits(rt, Rb, G, transposed, splits=0):
    """Binary relation Rb (0 / 1) as a bitmap through skf_gemm_bits: P = Rb @ G (transposed=0, Rb is M x K) or
    Q = Rb^T @ G (transposed=1, Rb is K x M)."""
    rows, cols = Rb.shape
    ldb_bytes = (cols + 63) // 64 * 8
    rows_pad = (rows + 63) // 64 * 64
    bits = np.zeros((rows_pad, ldb_bytes), np.uint8)
    bits[:rows, :(cols + 7) // 8] = np.packbits(Rb.astype(bool), axis=1, bitorder='little')
    if transposed:
        K, M = rows, cols
        Kp = rows_pad
    else:
        M, K = rows, cols
        Kp = ldb_bytes * 8
    N = G.shape[1]
    Gt = np.zeros((N, Kp), np.uint16)
    Gt[:, :K] = nat.to_bf16_bits(G.T)
    a, b = rt.mem.from_host(bits), rt.mem.from_host(Gt)
    c = rt.mem.empty(M * N * 4)
    ws = rt.mem.empty(40 * M * N * 4 + 256)
    rt.call('skf_gemm_bits', a.ptr, ldb_bytes, b.ptr, Kp, c.ptr, N, M, N, Kp, 1 if transposed else 0, splits,
            ws.ptr, ws.nbytes, None)
    got = rt.mem.to_host(c, (M, N), np.float32)
    Gr = nat.from_bf16_bits(Gt[:, :K]).astype(np.float64).T
    want = (Rb.T if transposed else Rb).astype(np.float64) @ Gr
    return got, want


@pytest.mark.parametrize('shape', [(1, 1, 1), (70, 100, 130), (128, 200, 257), (256, 129, 100), (200, 500, 200),
                                   (300, 64, 129), (100, 300, 520), (33, 260, 200)])
@pytest.mark.parametrize('transposed', [0, 1])
def test_gemm_bits_binary_relation_as_a_bitmap(rt, shape, transposed):
    """A 0 / 1 relation stored as one bit per entry, expanded to bf16 in LDS between the MFMA groups: both
    contractions (rows of the bitmap as output rows, or as the contraction index with transposed fragment
    reads), N <= 128 and N > 128 (3- and 2-deep G^T rings), tails everywhere, split-K."""
    N, rows, cols = shape
    rs = np.random.RandomState(rows + cols + N + transposed)
    Rb = (rs.rand(rows, cols) < 0.3).astype(np.float64)
    G = rs.randn(rows if transposed else cols, N)
    for splits in (0, 1, 2):
        got, want = run_gemm_bits(rt, Rb, G, transposed, splits)
        assert relerr(got, want) < 2e-6, splits


def test_to_bf16_and_transpose(rt):
    rs = np.random.RandomState(2)
    X = rs.randn(70, 45)
    for src_dtype, npd in ((nat.SKF_F64, np.float64), (nat.SKF_F32, np.float32)):
        src = rt.mem.from_host(X.astype(npd))
        dst = rt.mem.from_host(np.zeros((70, 64), np.uint16))
        rt.call('skf_to_bf16', dst.ptr, 64, src_dtype, src.ptr, 45, 70, 45, 0, None)
        got = rt.mem.to_host(dst, (70, 64), np.uint16)
        np.testing.assert_array_equal(got[:, :45], nat.to_bf16_bits(X.astype(npd)))
        assert (got[:, 45:] == 0).all()
        dstT = rt.mem.from_host(np.zeros((45, 128), np.uint16))
        rt.call('skf_to_bf16', dstT.ptr, 128, src_dtype, src.ptr, 45, 70, 45, 1, None)
        gotT = rt.mem.to_host(dstT, (45, 128), np.uint16)
        np.testing.assert_array_equal(gotT[:, :70], nat.to_bf16_bits(X.astype(npd)).T)
        assert (gotT[:, 70:] == 0).all()


def run_pinv(rt, dtype, A, route=None):
    npd = nat.NP_DTYPE[dtype]
    n = A.shape[0]
    need = C.c_size_t()
    rt.call('skf_pinv_sym_workspace_bytes', n, C.byref(need))
    a = rt.mem.from_host(np.ascontiguousarray(A, dtype=npd))
    k = rt.mem.empty(n * n * np.dtype(npd).itemsize)
    ws = rt.mem.empty(need.value)
    rt.call('skf_pinv_sym', dtype, a.ptr, n, k.ptr, n, n, ws.ptr, need.value, None)
    if route is not None:
        # the verdict word of the operator's workspace (skf_pinv_sym: three n_pad^2 f64 matrices, the eigenvalue row, then
        # int words: order, original order at +16, verdict at +32): 1 = an inverse written straight into K (fast path, or
        # the multi-workgroup deflation), 2 = the one-workgroup deflation, 0 = the eigen-solver
        npad = (n + 1) // 2 * 2
        mat = (npad * npad * 8 + 255) // 256 * 256
        off = 3 * mat + (npad * 8 + 255) // 256 * 256 + 32 * 4
        words = rt.mem.to_host(ws, (need.value // 4,), np.int32)
        route.append(int(words[off // 4]))
    return rt.mem.to_host(k, (n, n), npd)


@pytest.mark.parametrize('n', [1, 2, 5, 10, 31, 32, 33, 50, 70])
def test_pinv_full_rank_matches_scipy(rt, n, monkeypatch=None):
    """All routes: blocked Cholesky (default), plain Cholesky, Jacobi eigen-solver (forced)."""
    import os
    import scipy.linalg as spla
    rs = np.random.RandomState(n)
    G = rs.rand(4 * n + 3, n)
    A = G.T @ G
    want = spla.pinv(A)
    for env in ({}, {'SKF_CHOL_UNBLOCKED': '1'}, {'SKF_PINV_JACOBI': '1'}):
        if env.get('SKF_PINV_JACOBI') and n > 50:
            continue
        os.environ.update(env)
        try:
            got = run_pinv(rt, nat.SKF_F64, A)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert relerr(got, want) < 1e-9 * max(1.0, np.linalg.cond(A) * 1e-3), env
        assert relerr(A @ got @ A, A) < 1e-11


@pytest.mark.parametrize('n', [65, 96, 130, 200, 256])
def test_pinv_blocked_sweep_orders_65_to_256(rt, n):
    """Orders 65 .. 256 take the blocked sweep operator (sweep_inverse_kernel: K written by the kernel itself): against
    scipy and against the blocked Cholesky inverse it replaced (SKF_PINV_SWEEP=0), badly scaled columns included (the pivot
    test is relative to each pivot's own diagonal entry, as in the Cholesky kernels); a singular matrix is declined and
    lands on the deflation path with the same result as before."""
    import os
    import scipy.linalg as spla
    rs = np.random.RandomState(n)
    G = rs.rand(3 * n + 5, n)
    A = G.T @ G
    want = spla.pinv(A)
    got = run_pinv(rt, nat.SKF_F64, A)
    os.environ['SKF_PINV_SWEEP'] = '0'
    try:
        old = run_pinv(rt, nat.SKF_F64, A)
    finally:
        os.environ.pop('SKF_PINV_SWEEP', None)
    assert relerr(got, want) < 1e-9 * max(1.0, np.linalg.cond(A) * 1e-3)
    assert relerr(got, old) < 1e-9 * max(1.0, np.linalg.cond(A) * 1e-3)
    assert relerr(A @ got @ A, A) < 1e-11
    assert np.abs(got - got.T).max() <= 1e-12 * np.abs(got).max()
    if n in (96, 200):
        scale = 10.0 ** (-4.0 * rs.rand(n))
        Gs = rs.rand(4 * n, n) * scale
        As = Gs.T @ Gs
        d = np.sqrt(np.diag(As))
        gs = run_pinv(rt, nat.SKF_F64, As)
        # (cond(As) ~ 4e9: scipy's own inverse is off by 1e-8 there; the equilibrated matrix has cond ~ 2e3 and its inverse,
        # scaled back, is the reference -- the sweep is within 1e-13 of it, the Cholesky inverse within 3e-14)
        W = np.outer(d, d)
        assert relerr(gs * W, np.linalg.inv(As / W)) < 1e-12
        Gd = rs.rand(n // 2, n)                        # rank n / 2: the sweep meets a failed pivot and hands over
        Ad = Gd.T @ Gd
        assert relerr(run_pinv(rt, nat.SKF_F64, Ad), spla.pinv(Ad)) < 1e-7


@pytest.mark.parametrize('n', [65, 96, 130, 200, 256])
def test_pinv_sweep_one_launch_per_block_step_keeps_the_bits(rt, n):
    """sweep_step_kernel (round 5: one launch per block step, the rank-32 update of a step spread over row slabs, the matrix
    alternating between two copies) against sweep_inverse_kernel (one workgroup, in place): every element takes the same
    arithmetic in the same order, so the inverses are equal bit for bit -- at slabs of 32 and 64 rows, with a ragged last
    block, and for a matrix that fails a pivot in a late step (both hand over to the deflation)."""
    import os
    rs = np.random.RandomState(100 + n)
    G = rs.rand(3 * n + 5, n)
    A = G.T @ G
    Gd = np.concatenate([rs.rand(n - 7, n - 7), np.zeros((n - 7, 7))], axis=1)     # pivots fail in the LAST block
    Gd[:, -7:] = Gd[:, :7]
    Ad = Gd.T @ Gd

    def run(step_min, rows=None):
        os.environ['SKF_SWEEP_STEP_MIN'] = str(step_min)
        if rows:
            os.environ['SKF_SWEEP_ROWS'] = str(rows)
        try:
            return run_pinv(rt, nat.SKF_F64, A), run_pinv(rt, nat.SKF_F64, Ad)
        finally:
            os.environ.pop('SKF_SWEEP_STEP_MIN', None)
            os.environ.pop('SKF_SWEEP_ROWS', None)

    one, one_d = run(0)
    for rows in (32, 64):
        stepped, stepped_d = run(65, rows)
        assert np.array_equal(one, stepped), rows
        assert np.array_equal(one_d, stepped_d), rows
    assert relerr(A @ one @ A, A) < 1e-11


@pytest.mark.parametrize('n', [257, 300, 420])
def test_pinv_blocked_sweep_above_order_256(rt, n):
    """Orders above 256 (round 5): the step-per-launch sweep with the column operands of the update read from the matrix in
    memory (sweep_step_kernel<true>; the LDS holds the pivot rows and the slab's rows only) against scipy and against the
    blocked Cholesky inverse it replaces there (SKF_SWEEP_BIG=0); a matrix that fails a pivot is handed to the deflation."""
    import os
    import scipy.linalg as spla
    rs = np.random.RandomState(n)
    G = rs.rand(3 * n + 5, n)
    A = G.T @ G
    got = run_pinv(rt, nat.SKF_F64, A)
    os.environ['SKF_SWEEP_BIG'] = '0'
    try:
        old = run_pinv(rt, nat.SKF_F64, A)
    finally:
        os.environ.pop('SKF_SWEEP_BIG', None)
    want = spla.pinv(A)
    bound = 1e-9 * max(1.0, np.linalg.cond(A) * 1e-3)
    assert relerr(got, want) < bound and relerr(got, old) < bound
    assert relerr(A @ got @ A, A) < 1e-11
    assert np.abs(got - got.T).max() <= 1e-12 * np.abs(got).max()
    if n == 300:
        Gd = rs.rand(n - 40, n)                        # rank n - 40: a pivot of the last blocks fails
        Ad = Gd.T @ Gd
        assert relerr(run_pinv(rt, nat.SKF_F64, Ad), spla.pinv(Ad)) < 1e-7


def test_pinv_rank_deficient_truncates_like_scipy(rt):
    """reference tests/test_n_run.py:14: rank 50 factor of 30 objects -> Gram has rank 30."""
    import scipy.linalg as spla
    rs = np.random.RandomState(1)
    G = rs.rand(30, 50)
    A = G.T @ G
    got = run_pinv(rt, nat.SKF_F64, A)
    want = spla.pinv(A)
    assert np.abs(got).max() < 10 * np.abs(want).max()
    assert relerr(got, want) < 1e-7
    got32 = run_pinv(rt, nat.SKF_F32, A)
    assert np.isfinite(got32).all()


@pytest.mark.parametrize('n,rank', [(50, 30), (96, 40), (130, 65), (33, 1)])
def test_pinv_deflation_matches_scipy_and_the_eigen_path(rt, n, rank, monkeypatch):
    """A rank-deficient Gram matrix with a clear spectral gap goes through the rank-revealing deflation
    (pivoted Cholesky, A^+ = Y Y^T with Y = L (L^T L)^-1): same result as scipy.linalg.pinv and as the Jacobi
    eigen path with its exact cut-off (SKF_PINV_JACOBI=1), duplicate columns included."""
    import scipy.linalg as spla
    rs = np.random.RandomState(n + rank)
    G = rs.rand(rank, n)
    G[:, n // 2] = G[:, 1]                       # an exactly duplicated latent column
    A = G.T @ G
    want = spla.pinv(A)
    got = run_pinv(rt, nat.SKF_F64, A)
    assert relerr(got, want) < 1e-8
    monkeypatch.setenv('SKF_PINV_JACOBI', '1')
    exact = run_pinv(rt, nat.SKF_F64, A)
    assert relerr(got, exact) < 1e-8


@pytest.mark.parametrize('n,rank', [(300, 170), (420, 97), (260, 259)])
def test_pinv_deflation_over_several_workgroups_above_order_256(rt, n, rank, monkeypatch):
    """Round 6: above order 256 a rank-deficient Gram matrix is deflated by pchol_step_kernel -- one launch per block of
    up to 32 pivots over slabs of 64 rows, pivots rejected inside a block when the block's earlier pivots took them to the
    noise level (a duplicated column is chosen twice by the stale diagonal) --, B = L^T L is inverted by the blocked sweep of
    the fast path and K = Y Y^T leaves two gated products: same result as scipy.linalg.pinv and as the one-workgroup
    kernel (SKF_SWEEP_BIG=0 switches the multi-workgroup route off with the fast path it builds on)."""
    import scipy.linalg as spla
    rs = np.random.RandomState(n + rank)
    G = rs.rand(rank, n)
    G[:, n // 2] = G[:, 1]                       # an exactly duplicated latent column
    G[:, 7] = 0.0                                # and an all-zero one
    A = G.T @ G
    want = spla.pinv(A)
    route = []
    got = run_pinv(rt, nat.SKF_F64, A, route)
    assert route == [1]                          # written straight into K by the multi-workgroup route
    assert relerr(got, want) < 1e-8, relerr(got, want)
    assert relerr(A @ got @ A, A) < 1e-10
    assert np.abs(got - got.T).max() <= 1e-10 * np.abs(got).max()
    monkeypatch.setenv('SKF_SWEEP_BIG', '0')
    one_wg = run_pinv(rt, nat.SKF_F64, A, route)
    assert route == [1, 2]                       # ... and by the one-workgroup kernel (eigen format) without it
    assert relerr(got, one_wg) < 1e-8


def test_pinv_ambiguous_spectrum_above_order_256_is_left_to_the_exact_cut_off(rt):
    """The multi-workgroup deflation applies the same gap test: an accepted pivot inside (1e-10, 1e-7) of the largest
    declines the matrix, and the eigen path with scipy's cut-off takes it."""
    import scipy.linalg as spla
    rs = np.random.RandomState(5)
    n = 258
    Qm, _ = np.linalg.qr(rs.randn(n, n))
    w = np.array([1.0] * 40 + [1e-8] * 2 + [0.0] * (n - 42))
    A = (Qm * w) @ Qm.T
    A = 0.5 * (A + A.T)
    route = []
    got = run_pinv(rt, nat.SKF_F64, A, route)
    assert route == [0]                           # declined by both deflations: the eigen-solver
    want = spla.pinv(A)
    assert np.abs(got).max() > 1e6                # the 1e-8 directions were inverted, not dropped
    assert relerr(got, want) < 1e-6


def test_pinv_ambiguous_spectrum_falls_back_to_the_exact_cut_off(rt):
    """A singular value inside the deflation's gap band (1e-10 .. 1e-7 of the largest) is neither noise nor safely
    invertible by a pivot rule: the deflation declines and the eigen path applies scipy's cut-off (which keeps
    it: 1e-8 is far above n * eps)."""
    import scipy.linalg as spla
    rs = np.random.RandomState(4)
    Qm, _ = np.linalg.qr(rs.randn(24, 24))
    w = np.array([1.0] * 10 + [1e-8] * 2 + [0.0] * 12)
    A = (Qm * w) @ Qm.T
    A = 0.5 * (A + A.T)
    got = run_pinv(rt, nat.SKF_F64, A)
    want = spla.pinv(A)
    assert np.abs(got).max() > 1e7                # the 1e-8 directions were inverted, not dropped
    assert relerr(got, want) < 1e-6


@pytest.mark.parametrize('n', [12, 40, 70])
def test_pinv_badly_scaled_columns_match_scipy(rt, n):
    """Latent dimensions of very different scale (column norms down to 1e-5 of the largest: diagonal of
    the Gram matrix spread over 1e-10) but independent: the pivot test of the Cholesky fast path is
    relative to each pivot's own diagonal, so these stay on it; scipy inverts them too (sigma_min /
    sigma_max ~ 1e-10 is far above its cut-off).  Compared in the equilibrated metric."""
    import os
    import scipy.linalg as spla
    rs = np.random.RandomState(n)
    scale = 10.0 ** (-5.0 * rs.rand(n))
    scale[0], scale[-1] = 1.0, 1e-5
    G = rs.rand(6 * n, n) * scale
    A = G.T @ G
    want = spla.pinv(A)
    d = np.sqrt(np.diag(A))
    for env in ({}, {'SKF_CHOL_NO_SMALL': '1'}, {'SKF_CHOL_UNBLOCKED': '1'}):
        os.environ.update(env)
        try:
            got = run_pinv(rt, nat.SKF_F64, A)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert relerr(got * np.outer(d, d), want * np.outer(d, d)) < 1e-8, env


def test_pinv_zero_and_diagonal(rt):
    got = run_pinv(rt, nat.SKF_F64, np.zeros((6, 6)))
    assert (got == 0).all()
    d = np.diag([4.0, 2.0, 0.0, 1e-30, 5.0])
    got = run_pinv(rt, nat.SKF_F64, d)
    np.testing.assert_allclose(got, np.diag([0.25, 0.5, 0.0, 0.0, 0.2]), atol=1e-15)


def test_fill_uniform_matches_oracle_hash(rt):
    from oracle.dfmf_oracle import hash_uniform_matrix
    for dtype, npd in ((nat.SKF_F64, np.float64), (nat.SKF_F32, np.float32)):
        buf = rt.mem.empty(37 * 53 * 8)
        rt.call('skf_fill_uniform', dtype, buf.ptr, 37, 53, 53, 9, 1.0, 0.0, None)
        got = rt.mem.to_host(buf, (37, 53), npd)
        np.testing.assert_array_equal(got, hash_uniform_matrix(9, 37, 53).astype(npd))
    buf = rt.mem.empty(8 * 16 * 2)
    rt.call('skf_fill_uniform', nat.SKF_BF16, buf.ptr, 8, 16, 16, 3, 2.0, -1.0, None)
    got = rt.mem.to_host(buf, (8, 16), np.uint16)
    want = (hash_uniform_matrix(3, 8, 16) * 2.0 - 1.0).astype(np.float32)
    back = (got.astype(np.uint32) << 16).view(np.float32)
    assert np.abs(back - want).max() <= np.abs(want).max() * 2 ** -8


def test_errors_are_reported_not_thrown(rt):
    lib = rt.lib
    assert lib.skf_gemm(nat.SKF_F32, 0, None, None, 0, None) == -1
    assert b'null' in lib.skf_last_error()
    with pytest.raises(nat.SkfNativeError):
        rt.call('skf_pinv_sym', nat.SKF_F64, None, 1, None, 1, 4, None, 0, None)
    out = nat._P()
    t = (nat.TypeDesc * 1)()
    t[0].n_obj, t[0].rank = 5, 0
    opt = nat.Options(nat.SKF_F64, nat.SKF_DFMF, -1, 0)
    assert lib.skf_plan_create(1, t, 0, None, 0, None, C.byref(opt), C.byref(out)) == -1
