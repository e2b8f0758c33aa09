// skf_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the DFMF/DFMC update loop.
//
// Data layout in HBM: every matrix is row-major with an explicit leading dimension.
// Masters (G, E, D, S, Gram, K, P, Q) are f64 in the SKF_F64 engine and f32 otherwise.
//
// Kernels
//   gemm_mfma_kernel<T,..>   C = epi(aop(A) * B)   tall-skinny / small dense contractions on the
//                            matrix cores: v_mfma_f32_32x32x2_f32 (T=float, exact f32, 157 TF
//                            peak) and v_mfma_f64_16x16x4_f64 (T=double).  Operands are addressed
//                            through (row,col) strides so that R, R^T, G, G^T, S, S^T all feed
//                            the same kernel; tiles are staged k-major in LDS so that the MFMA
//                            fragment reads are conflict-free ds_read_b32 / b64.
//   gemm_valu_kernel<T>      the same contract on the vector ALU (bring-up / cross-check engine).
//   splitk_reduce_kernel<T>  deterministic second stage for split-K launches + the epilogues.
//   jacobi_eigh_kernel       symmetric c x c eigen-decomposition (parallel two-sided cyclic
//                            Jacobi, f64, one workgroup per matrix) -> pinv with the SVD cut-off
//                            of scipy.linalg.pinv (reference _dfmf.py:232).
//   mult_update_kernel<T>    G <- G * sqrt(E / max(D, eps))               (_dfmf.py:294-296)
//   fill_uniform_kernel<T>   counter-based synthetic data, identical to oracle hash_uniform().
//   mask / cast / sqerr helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "skf_asm.h"      // LDS / global instructions in inline-assembly form, with counted waits (gfx950)

namespace skf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum AOp { AOP_NONE = 0, AOP_POS = 1, AOP_NEG = 2 };                 // x, max(x,0), max(-x,0)
enum Epi {
    EPI_STORE = 0,         // C  = v
    EPI_ACC = 1,           // C += v
    EPI_SPLIT_STORE = 2,   // C  = max(v,0) ; C2  = max(-v,0)
    EPI_SPLIT_ACC = 3,     // C += max(v,0) ; C2 += max(-v,0)       (_dfmf.py:256-258,278-282)
    EPI_MASKED_STORE = 4,  // C  = v where mask != 0                (_dfmc.py:319-325)
    EPI_SQDIFF = 5,        // per-workgroup partial of sum (C - v)^2 -> C2[block]  (C untouched)
    EPI_STORE_F32 = 6      // C = v (f64) ; C2 = (float)v: a backbone and its f32 rounding for the f32 engines' side products
};

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    void* C2;
    const uint8_t* mask;     // EPI_MASKED_STORE: same shape as C; one byte per entry (leading dim ldmask), or -- mask_bits
                             // -- one BIT per entry: bit (n & 7) of byte mask[m * ldmask + (n >> 3)], ldmask in bytes
    void* part;              // split-K partials [gridDim.z][M][N] (used when gridDim.z > 1)
    int64_t sa_m, sa_k;      // A(m,k) = A[m*sa_m + k*sa_k]
    int64_t sb_k, sb_n;      // B(k,n) = B[k*sb_k + n*sb_n]
    int64_t ldc, ldc2, ldmask;
    int M, N, K;
    int k_chunk;             // K range handled by one z-slice
    int aop, epi, nan_to_num;
    int c_bf16;              // EPI_SQDIFF: the matrix behind C is stored as bf16
    int mask_bits;           // EPI_MASKED_STORE: the mask is packed (the engine's own masks always are)
    const int* gate;         // when set: the launch does nothing unless *gate != 0 (a verdict earlier launches left on the device:
                             // the finishing products of the multi-workgroup deflation, which the host issues blind)
    int sym;                 // split-K launches of a SYMMETRIC product (Gram = G^T G): bm | bn << 16 of the launch's tile -- the
                             // grid lists only the tiles on / below the diagonal (gridDim.x = their number, gridDim.y = 1) and
                             // the reduce takes an element of a tile that was not computed from its mirror image (element
                             // (a, b) and (b, a) are the same products added in the same order: bit for bit the full
                             // product, 6 of 8 tiles at order 256); 0 = off
};

// the partial-sum element the reduce of a split-K launch reads for output element e = (m, n) (see GemmArgs::sym)
__device__ __forceinline__ int64_t sym_source(const GemmArgs& g, int64_t e) {
    if (!g.sym) return e;
    const int bm = g.sym & 0xffff, bn = g.sym >> 16;
    const int m = (int)(e / g.N), n = (int)(e % g.N);
    return ((n / bn) * bn >= (m / bm) * bm + bm) ? (int64_t)n * g.N + m : e;
}

__device__ __forceinline__ bool mask_test(const GemmArgs& g, int m, int n) {
    if (g.mask_bits) return (g.mask[(int64_t)m * g.ldmask + (n >> 3)] >> (n & 7)) & 1;
    return g.mask[(int64_t)m * g.ldmask + n] != 0;
}

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
template <typename T> struct Lim;
template <> struct Lim<float> { static __device__ __host__ float big() { return 3.40282346638528859812e+38f; } };
template <> struct Lim<double> { static __device__ __host__ double big() { return 1.79769313486231570815e+308; } };

__device__ __host__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    union { float f; uint32_t u; } x;
    x.f = f;
    if ((x.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x.u >> 16) | 0x40);   // quiet NaN
    const uint32_t r = 0x7fffu + ((x.u >> 16) & 1u);
    return (uint16_t)((x.u + r) >> 16);
}

__device__ __host__ __forceinline__ float bf16_to_f32(uint16_t h) {
    union { uint32_t u; float f; } x;
    x.u = (uint32_t)h << 16;
    return x.f;
}

// numpy.nan_to_num: NaN -> 0, +inf -> largest finite, -inf -> most negative finite
template <typename T>
__device__ __forceinline__ T nan_to_num(T x) {
    if (x != x) return (T)0;
    if (x > Lim<T>::big()) return Lim<T>::big();
    if (x < -Lim<T>::big()) return -Lim<T>::big();
    return x;
}

template <typename T>
__device__ __forceinline__ T apply_aop(T x, int aop) {
    if (aop == AOP_POS) return x > (T)0 ? x : (T)0;
    if (aop == AOP_NEG) return x > (T)0 ? (T)0 : -x;      // (t-1)*x of the reference, >= 0
    return x;
}

template <typename T>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int m, int n, T v) {
    T* C = (T*)g.C;
    T* C2 = (T*)g.C2;
    if (g.nan_to_num) v = nan_to_num(v);
    const int64_t i = (int64_t)m * g.ldc + n;
    const int64_t i2 = (int64_t)m * g.ldc2 + n;
    switch (g.epi) {
        case EPI_STORE: C[i] = v; break;
        case EPI_ACC: C[i] += v; break;
        case EPI_SPLIT_STORE:
            C[i] = v > (T)0 ? v : (T)0;
            C2[i2] = v > (T)0 ? (T)0 : -v;
            break;
        case EPI_SPLIT_ACC:
            C[i] += v > (T)0 ? v : (T)0;
            C2[i2] += v > (T)0 ? (T)0 : -v;
            break;
        case EPI_MASKED_STORE:
            if (mask_test(g, m, n)) C[i] = v;
            break;
        case EPI_STORE_F32:
            C[i] = v;
            ((float*)g.C2)[i2] = (float)v;
            break;
        default: break;
    }
}

// the accumulating epilogues with the old values handed in: a tile's old values are loaded together BEFORE its first store
// (element by element, `C[i] += v` is a memory round trip per element -- the compiler cannot move the next load above a
// store that may alias it; measured on the fused side update: 14-29 us of the 25-37 a workgroup took)
template <typename T>
__device__ __forceinline__ void epilogue_store_acc(const GemmArgs& g, int m, int n, T v, T old1, T old2) {
    T* C = (T*)g.C;
    T* C2 = (T*)g.C2;
    if (g.nan_to_num) v = nan_to_num(v);
    const int64_t i = (int64_t)m * g.ldc + n;
    const int64_t i2 = (int64_t)m * g.ldc2 + n;
    if (g.epi == EPI_ACC) {
        C[i] = old1 + v;
    } else {                                 // EPI_SPLIT_ACC
        C[i] = old1 + (v > (T)0 ? v : (T)0);
        C2[i2] = old2 + (v > (T)0 ? (T)0 : -v);
    }
}

// sum over the 64 lanes of a wave (result valid in every lane)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// MFMA traits: one matrix-core instruction per (T): tile MT x NT, depth KT per instruction
// ------------------------------------------------------------------------------------------
template <typename T> struct Mfma;

template <> struct Mfma<float> {          // v_mfma_f32_32x32x2_f32: 16 accumulator regs / lane
    static constexpr int MT = 32, NT = 32, KT = 2, NREG = 16;
    typedef f32x16 acc_t;
    static __device__ __forceinline__ int a_row(int lane) { return lane & 31; }
    static __device__ __forceinline__ int ab_k(int lane) { return lane >> 5; }
    static __device__ __forceinline__ int d_row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
    static __device__ __forceinline__ int d_col(int lane) { return lane & 31; }
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

template <> struct Mfma<double> {         // v_mfma_f64_16x16x4_f64: 4 f64 accumulators / lane
    static constexpr int MT = 16, NT = 16, KT = 4, NREG = 4;
    typedef f64x4 acc_t;
    static __device__ __forceinline__ int a_row(int lane) { return lane & 15; }
    static __device__ __forceinline__ int ab_k(int lane) { return lane >> 4; }
    static __device__ __forceinline__ int d_row(int lane, int r) { return (lane >> 4) + 4 * r; }
    static __device__ __forceinline__ int d_col(int lane) { return lane & 15; }
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
};

// ------------------------------------------------------------------------------------------
// cooperative tile staging: global (any strides) -> registers -> LDS (k-major, padded)
// 256 threads; element e = tid + i*256.  The fast index follows the contiguous global stride
// so that a wave touches whole 128-256 B segments.
// ------------------------------------------------------------------------------------------
constexpr int GEMM_THREADS = 256;

// stage_load only issues loads (indices clamped into the matrix, raw storage type, no use of
// the loaded value -> no wait, no divergent branch); conversion, the Theta+/- operand op and the
// zero fill of the K tail happen in stage_store, after the MFMA work that hides the latency.
// Rows past the end need no zero fill: they only feed accumulator rows that are never stored.
//
// mode (wave-uniform, chosen once per launch by stage_mode):
//   STAGE_SCALAR  one element per load (any strides, any alignment)
//   STAGE_VEC_K   16-byte loads along a K-contiguous operand   (element (r, k..k+V-1))
//   STAGE_VEC_R   16-byte loads along a row-contiguous operand (element (r..r+V-1, k))
enum { STAGE_SCALAR = 0, STAGE_VEC_K = 1, STAGE_VEC_R = 2 };

template <typename TS>
__device__ __forceinline__ int stage_mode(const TS* src, int64_t s_row, int64_t s_k, int row_end, int k_lo, int k_end) {
    constexpr int V = 16 / (int)sizeof(TS);
    const bool aligned = (((uintptr_t)src) & 15) == 0;
    if (s_k == 1 && aligned && s_row % V == 0 && k_end % V == 0 && k_lo % V == 0) return STAGE_VEC_K;
    if (s_row == 1 && aligned && s_k % V == 0 && row_end % V == 0) return STAGE_VEC_R;
    return STAGE_SCALAR;
}

template <typename TS, int ROWS, int BK>
__device__ __forceinline__ void stage_load(TS (&reg)[ROWS * BK / GEMM_THREADS], const TS* __restrict__ src,
                                           int64_t s_row, int64_t s_k, int row0, int k0, int row_end,
                                           int k_end, int tid, int mode = STAGE_SCALAR) {
    constexpr int PER = ROWS * BK / GEMM_THREADS;
    constexpr int V = 16 / (int)sizeof(TS);
    union Vec { u32x4 v; TS t[V]; };
    if (PER % V != 0) mode = STAGE_SCALAR;          // tile too small for whole vectors per thread
    if (mode == STAGE_VEC_K) {
#pragma unroll
        for (int i = 0; i < PER / V; ++i) {
            const int e = tid + i * GEMM_THREADS;
            const int k = (e % (BK / V)) * V;
            const int r = e / (BK / V);
            int gr = row0 + r, gk = k0 + k;
            gr = gr < row_end ? gr : row_end - 1;
            gk = gk < k_end ? gk : k_end - V;               // K tail: a whole in-bounds vector
            Vec u;
            u.v = *(const u32x4*)(src + (int64_t)gr * s_row + gk);
#pragma unroll
            for (int j = 0; j < V; ++j) reg[i * V + j] = u.t[j];
        }
    } else if (mode == STAGE_VEC_R) {
#pragma unroll
        for (int i = 0; i < PER / V; ++i) {
            const int e = tid + i * GEMM_THREADS;
            const int r = (e % (ROWS / V)) * V;
            const int k = e / (ROWS / V);
            int gr = row0 + r, gk = k0 + k;
            gr = gr < row_end ? gr : row_end - V;
            gk = gk < k_end ? gk : k_end - 1;
            Vec u;
            u.v = *(const u32x4*)(src + gr + (int64_t)gk * s_k);
#pragma unroll
            for (int j = 0; j < V; ++j) reg[i * V + j] = u.t[j];
        }
    } else {
        const bool k_fast = (s_k == 1);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * GEMM_THREADS;
            const int k = k_fast ? (e % BK) : (e / ROWS);
            const int r = k_fast ? (e / BK) : (e % ROWS);
            int gr = row0 + r, gk = k0 + k;
            gr = gr < row_end ? gr : row_end - 1;
            gk = gk < k_end ? gk : k_end - 1;
            reg[i] = src[(int64_t)gr * s_row + (int64_t)gk * s_k];
        }
    }
}

// FULL: the K tile lies inside [k_lo, k_end) as a whole -- no zero fill of a K tail (the test per element costs a compare
// and a select each, and with a run-time `aop` beside it the compiler left a chain of scalar branches PER ELEMENT in the
// store phase of the f32 contraction: callers pass the literal AOP_NONE on the hot path).
template <typename T, typename TS, int ROWS, int BK, int LD, bool FULL = false>
__device__ __forceinline__ void stage_store(T (*lds)[LD], const TS (&reg)[ROWS * BK / GEMM_THREADS],
                                            bool k_fast, int k0, int k_end, int aop, int tid,
                                            int mode = STAGE_SCALAR) {
    constexpr int PER = ROWS * BK / GEMM_THREADS;
    constexpr int V = 16 / (int)sizeof(TS);
    if (PER % V != 0) mode = STAGE_SCALAR;
    if (mode == STAGE_VEC_K) {
#pragma unroll
        for (int i = 0; i < PER / V; ++i) {
            const int e = tid + i * GEMM_THREADS;
            const int k = (e % (BK / V)) * V;
            const int r = e / (BK / V);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                T v = apply_aop((T)reg[i * V + j], aop);
                lds[k + j][r] = (FULL || k0 + k + j < k_end) ? v : (T)0;
            }
        }
    } else if (mode == STAGE_VEC_R) {
#pragma unroll
        for (int i = 0; i < PER / V; ++i) {
            const int e = tid + i * GEMM_THREADS;
            const int r = (e % (ROWS / V)) * V;
            const int k = e / (ROWS / V);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                T v = apply_aop((T)reg[i * V + j], aop);
                lds[k][r + j] = (FULL || k0 + k < k_end) ? v : (T)0;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * GEMM_THREADS;
            const int k = k_fast ? (e % BK) : (e / ROWS);
            const int r = k_fast ? (e / BK) : (e % ROWS);
            T v = apply_aop((T)reg[i], aop);
            lds[k][r] = (FULL || k0 + k < k_end) ? v : (T)0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// MFMA GEMM.  Workgroup = 4 waves in a 2 x 2 arrangement; wave tile = (WR*MT) x (WC*NT);
// block tile BM x BN = 2*WR*MT x 2*WC*NT; K tile BK; grid = (ceil(N/BN), ceil(M/BM), splits).
// One K step:  [global loads of tile t+1 in flight]  MFMA on tile t from LDS  | barrier |
//              registers -> LDS | barrier.
// ------------------------------------------------------------------------------------------
// TAG 0 / 1 only change the symbol name: TAG=1 instantiations are the two relation contractions
// P = R G_j and Q = R^T G_i (the only launches that read R), so that profilers list the
// dominant kernel separately from the small n x c x c products that share the code.
// T = arithmetic / output type (f32 or f64 MFMA); TA, TB = storage types of the operands (a f32
// operand feeding an f64 contraction is widened while it is staged: Gram / G^T P in the f32
// engine; an f64 backbone feeding an f32 contraction is narrowed the same way).
// (f32: at least 2 waves per SIMD, i.e. <= 256 registers per lane -- left alone the compiler
// spreads the unrolled staging code over 277 registers and halves the occupancy)
// FM >= 0: compile-time staging modes (bits 0-1 operand A, 2-3 operand B), conditions checked by the host.
template <typename T, typename TA, typename TB, int WR, int WC, int BK, int TAG, int FM = -1>
__device__ __forceinline__ void gemm_mfma_body(const GemmArgs& g, const unsigned bid_x, const unsigned bid_y, const unsigned bid_z,
                                               const unsigned grid_x, const unsigned grid_z) {
    typedef Mfma<T> MF;
    constexpr int BM = 2 * WR * MF::MT, BN = 2 * WC * MF::NT;
    // Row pitch of the k-major LDS tiles.  The fragment reads (consecutive lanes, consecutive words) are conflict-free at any
    // pitch; what matters is how the operand is WRITTEN.  A K-contiguous operand is stored transposed, one word per store and
    // four stores a pitch apart: ds_write_b32 serves lanes 0-31 / 32-63 in one cycle each on 32 banks, and at a pitch of
    // BM + 1 words the 8 k-columns x 4 rows of a lane group land on 32 different banks.  A row-contiguous operand is stored
    // as four consecutive words per lane: as four ds_write_b32 they hit every bank four times (lanes l and l + 8 of a group
    // are 32 words apart -- PMC, round 5: a third of the LDS cycles of the f32 contraction were bank conflicts); at a pitch
    // of BM + 4 every row starts 16-byte aligned and the four words go out as ONE ds_write_b128 (conflicts: 9.4e8 -> 3e7
    // cycles per launch).  f64 tiles keep their pitch (their products sit at the f64 matrix-core rate either way).
    constexpr int MA = FM >= 0 ? (FM & 3) : -1, MB = FM >= 0 ? ((FM >> 2) & 3) : -1;
    constexpr int PADA = (sizeof(T) == 4 && MA == STAGE_VEC_R) ? 4 : 1;
    constexpr int PADB = (sizeof(T) == 4 && MB == STAGE_VEC_R) ? 4 : 1;
    constexpr int LDA = BM + PADA, LDB = BN + PADB;
    // (Round 5, measured and not kept: two LDS images of the tiles with one barrier per K step and the fragments of the next
    // step read ahead of the products -- f32 P12 88 instead of 92 TFLOP/s, f64 engine 5.73 instead of 5.91 it/s.)
    __shared__ __attribute__((aligned(16))) T As[BK][LDA];
    __shared__ __attribute__((aligned(16))) T Bs[BK][LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (WR * MF::MT), wn0 = (wave & 1) * (WC * MF::NT);
    int bm0 = bid_y * BM, bn0 = bid_x * BN;
    if (g.sym) {
        // symmetric product: the launch lists only the tiles on / below the diagonal, blockIdx.x = their running number
        // row tile by row tile (a launch of ALL tiles whose upper ones return at once leaves whole XCDs idle: workgroups go
        // to the XCDs round robin, and with 2 x 4 tiles the skipped ones are always the same residues mod 8)
        int left = (int)bid_x, by = 0;
        for (;; ++by) {
            int cnt = (by * BM + BM - 1) / BN + 1;                // column tiles of row tile `by` that touch the lower triangle
            const int all = (g.N + BN - 1) / BN;
            cnt = cnt < all ? cnt : all;
            if (left < cnt) break;
            left -= cnt;
        }
        bm0 = by * BM;
        bn0 = left * BN;
    }
    const int kz0 = bid_z * g.k_chunk;
    const int kz1 = (kz0 + g.k_chunk < g.K) ? kz0 + g.k_chunk : g.K;
    const TA* __restrict__ A = (const TA*)g.A;
    const TB* __restrict__ B = (const TB*)g.B;
    const bool a_kfast = (g.sa_k == 1), b_kfast = (g.sb_k == 1);

    typename MF::acc_t acc[WR][WC];
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) acc[i][j][r] = (T)0;

    TA ra[BM * BK / GEMM_THREADS];
    TB rb[BN * BK / GEMM_THREADS];
    const int nkt = (kz1 - kz0 + BK - 1) / BK;
    const int ma = FM >= 0 ? (FM & 3) : stage_mode<TA>(A, g.sa_m, g.sa_k, g.M, kz0, kz1);
    const int mb = FM >= 0 ? ((FM >> 2) & 3) : stage_mode<TB>(B, g.sb_n, g.sb_k, g.N, kz0, kz1);
    // registers -> LDS of the tile that starts at k0.  The common case -- a whole K tile, no operand op -- is its own
    // instantiation with compile-time constants (uniform branches, taken once per tile, not once per element).
    auto store_tiles = [&](int k0) {
        if (k0 + BK <= kz1 && g.aop == AOP_NONE) {
            stage_store<T, TA, BM, BK, LDA, true>(As, ra, a_kfast, k0, kz1, AOP_NONE, tid, ma);
            stage_store<T, TB, BN, BK, LDB, true>(Bs, rb, b_kfast, k0, kz1, AOP_NONE, tid, mb);
        } else {
            stage_store<T, TA, BM, BK, LDA>(As, ra, a_kfast, k0, kz1, g.aop, tid, ma);
            stage_store<T, TB, BN, BK, LDB>(Bs, rb, b_kfast, k0, kz1, AOP_NONE, tid, mb);
        }
    };
    if (nkt > 0) {
        stage_load<TA, BM, BK>(ra, A, g.sa_m, g.sa_k, bm0, kz0, g.M, kz1, tid, ma);
        stage_load<TB, BN, BK>(rb, B, g.sb_n, g.sb_k, bn0, kz0, g.N, kz1, tid, mb);
        store_tiles(kz0);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1 < nkt);
        const int k_next = kz0 + (kt + 1) * BK;
        if (more) {
            stage_load<TA, BM, BK>(ra, A, g.sa_m, g.sa_k, bm0, k_next, g.M, kz1, tid, ma);
            stage_load<TB, BN, BK>(rb, B, g.sb_n, g.sb_k, bn0, k_next, g.N, kz1, tid, mb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += MF::KT) {
            T a[WR], b[WC];
            const int kr = kk + MF::ab_k(lane);
#pragma unroll
            for (int i = 0; i < WR; ++i) a[i] = As[kr][wm0 + i * MF::MT + MF::a_row(lane)];
#pragma unroll
            for (int j = 0; j < WC; ++j) b[j] = Bs[kr][wn0 + j * MF::NT + MF::a_row(lane)];
#pragma unroll
            for (int i = 0; i < WR; ++i)
#pragma unroll
                for (int j = 0; j < WC; ++j) acc[i][j] = MF::mma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            store_tiles(k_next);
        }
        __syncthreads();
    }

    // ---- epilogue
    const bool split = (grid_z > 1);
    T sq = (T)0;
    if (!split && (g.epi == EPI_ACC || g.epi == EPI_SPLIT_ACC)) {          // (uniform)
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
            for (int j = 0; j < WC; ++j) {
                T o1[MF::NREG], o2[MF::NREG];
                const int n = bn0 + wn0 + j * MF::NT + MF::d_col(lane);
#pragma unroll
                for (int r = 0; r < MF::NREG; ++r) {
                    const int m = bm0 + wm0 + i * MF::MT + MF::d_row(lane, r);
                    const bool in = m < g.M && n < g.N;
                    o1[r] = in ? ((const T*)g.C)[(int64_t)m * g.ldc + n] : (T)0;
                    o2[r] = (in && g.epi == EPI_SPLIT_ACC) ? ((const T*)g.C2)[(int64_t)m * g.ldc2 + n] : (T)0;
                }
#pragma unroll
                for (int r = 0; r < MF::NREG; ++r) {
                    const int m = bm0 + wm0 + i * MF::MT + MF::d_row(lane, r);
                    if (m < g.M && n < g.N) epilogue_store_acc<T>(g, m, n, acc[i][j][r], o1[r], o2[r]);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int m = bm0 + wm0 + i * MF::MT + MF::d_row(lane, r);
                const int n = bn0 + wn0 + j * MF::NT + MF::d_col(lane);
                if (m < g.M && n < g.N) {
                    const T v = acc[i][j][r];
                    if (split) {
                        ((T*)g.part)[((int64_t)bid_z * g.M + m) * g.N + n] = v;
                    } else if (g.epi == EPI_SQDIFF) {
                        const int64_t ci = (int64_t)m * g.ldc + n;
                        const T rv = g.c_bf16 ? (T)bf16_to_f32(((const uint16_t*)g.C)[ci]) : ((const T*)g.C)[ci];
                        const T d = rv - v;
                        sq += d * d;
                    } else {
                        epilogue_store<T>(g, m, n, v);
                    }
                }
            }
    if (!split && g.epi == EPI_SQDIFF) {
        __shared__ T red[GEMM_THREADS / 64];
        sq = wave_sum(sq);
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) {
            T s = (T)0;
            for (int w = 0; w < GEMM_THREADS / 64; ++w) s += red[w];
            ((T*)g.C2)[bid_y * grid_x + bid_x] = s;
        }
    }
}

template <typename T, typename TA, typename TB, int WR, int WC, int BK, int TAG, int FM = -1>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_mfma_kernel(GemmArgs g) {
    if (g.gate != nullptr && *g.gate == 0) return;                         // (uniform)
    gemm_mfma_body<T, TA, TB, WR, WC, BK, TAG, FM>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z);
}

// TWO independent unsplit products in one launch (blockIdx.z = which; round 5): the c x c chains of a relation hold pairs of
// products that do not depend on each other -- S Gram_j beside Gram_i S, U S^T beside S^T U' -- and a launch on the second
// stream costs its ~10 us whatever it computes.  The grid covers the larger tile count; a workgroup without a tile in its
// product returns.  Element by element the arithmetic of gemm_mfma_kernel.
template <typename T, typename TA, typename TB, int WR, int WC, int BK, int TAG, int FM = -1>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_mfma_pair_kernel(GemmArgs g0, GemmArgs g1) {
    typedef Mfma<T> MF;
    constexpr int BM = 2 * WR * MF::MT, BN = 2 * WC * MF::NT;
    const GemmArgs g = blockIdx.z ? g1 : g0;
    if ((int)(blockIdx.y * BM) >= g.M || (int)(blockIdx.x * BN) >= g.N) return;
    gemm_mfma_body<T, TA, TB, WR, WC, BK, TAG, FM>(g, blockIdx.x, blockIdx.y, 0u, gridDim.x, 1u);
}

// Up to FOUR independent split-K products of ONE instantiation in one launch (blockIdx.y = which; round 6): the Gram matrices
// G^T G of all types open an iteration as three launches of ~500 workgroups each, every one with its own ramp and tail and a
// reduce launch behind it.  Here they share one grid (x = the largest tile count, z = the largest slice count; a workgroup
// outside its product's range returns) and one reduce launch.  Each product keeps its own tile list, K slices and scratch
// region: element by element, slice by slice the arithmetic of gemm_mfma_kernel + splitk_reduce_z16_kernel.
struct GemmGroup {
    GemmArgs g[4];
    int tiles[4];        // gridDim.x of the product's own launch (GemmArgs::sym: the tiles on / below the diagonal)
    int splits[4];       // its K slices
};
template <typename T, typename TA, typename TB, int WR, int WC, int BK, int TAG, int FM = -1>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_mfma_group_kernel(GemmGroup m) {
    const unsigned w = blockIdx.y;
    const unsigned tiles = (unsigned)m.tiles[w], splits = (unsigned)m.splits[w];
    if (blockIdx.x >= tiles || blockIdx.z >= splits) return;
    gemm_mfma_body<T, TA, TB, WR, WC, BK, TAG, FM>(m.g[w], blockIdx.x, 0u, blockIdx.z, tiles, splits);
}

// ------------------------------------------------------------------------------------------
// Fused accumulator update of ONE relation side (reference _dfmf.py:254-264 + 278-279, or
// :266-276 + 281-282 for the column side):
//     A = X * Sop            X = P (n x k1) with Sop = S^T,  or  X = Q with Sop = S
//     E (+)= max(A,0) + G * Bn        D (+)= max(-A,0) + G * Bp
// in one pass over E / D: three MFMA contractions share one output tile.  accE first holds A,
// is split in registers into (A+, A-) and then both halves keep accumulating G*Bn / G*Bp (the
// MFMA C-operand), so E and D are written exactly once (read only when `accumulate`).
// X, G, E, D are T (the master type); Sop, Bn, Bp are TB (f64 c x c matrices).
// ------------------------------------------------------------------------------------------
struct SideArgs {
    const void* X;      // [n][ldx], k1 columns
    const void* Sop;    // B(k, j) = Sop[k*ss_k + j*ss_n], k < k1, j < c
    const void* G;      // [n][ldg], c columns
    const void* Bn;     // [c][ldb]
    const void* Bp;     // [c][ldb]
    void* E;            // [n][lde]
    void* D;
    int64_t ldx, ss_k, ss_n, ldg, ldb, lde;
    int n, c, k1;
    int accumulate;     // 0: E/D are overwritten (first contribution of the iteration)
    int nan_to_num;     // numpy.nan_to_num on A before the split (DFMF only)
    int phase2;         // 0: only the (X Sop) split terms; 1: also + G Bn / + G Bp   (k1 = 0: only those)
};

// FM >= 0: the staging modes are compile-time constants (bits 0-1 X, 2-3 Sop, 4-5 G, 6-7 Bn/Bp; the
// host checked the alignment conditions of stage_mode): one staging path per operand instead of
// three keeps the hoisted address arithmetic out of the register budget.  FM = -1: run-time modes.
#define SKF_SIDE_FM(mx, ms, mg, mb) ((mx) | ((ms) << 2) | ((mg) << 4) | ((mb) << 6))
template <typename T, typename TB, int WR, int WC, int BK, int FM = -1>
__global__ __launch_bounds__(GEMM_THREADS, (sizeof(T) == 4 ? 2 : 1)) void side_update_kernel(SideArgs a) {
    typedef Mfma<T> MF;
    constexpr int BM = 2 * WR * MF::MT, BN = 2 * WC * MF::NT;
    // (pitch of the B images: the c x c operands Bn / Bp are row-contiguous -- one ds_write_b128 per lane at a 16-byte aligned
    // pitch instead of four 4-way conflicting ds_write_b32, see gemm_mfma_kernel; f32 tiles with compile-time modes only)
    constexpr int LDA = BM + 1, LDB = BN + ((sizeof(T) == 4 && FM >= 0 && ((FM >> 6) & 3) == STAGE_VEC_R) ? 4 : 1);
    __shared__ __attribute__((aligned(16))) T As[BK][LDA];
    __shared__ __attribute__((aligned(16))) T Bs[BK][LDB];
    __shared__ __attribute__((aligned(16))) T Bs2[BK][LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (WR * MF::MT), wn0 = (wave & 1) * (WC * MF::NT);
    const int bm0 = blockIdx.y * BM, bn0 = blockIdx.x * BN;

    typename MF::acc_t accE[WR][WC], accD[WR][WC];
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) accE[i][j][r] = (T)0;

    T ra[BM * BK / GEMM_THREADS];
    TB rb[BN * BK / GEMM_THREADS], rb2[BN * BK / GEMM_THREADS];
#ifdef SKF_PROBE_STAMPS
    const long long su0 = wall_clock64();
#endif

    // ---- phase 1: accE = X * Sop
    {
        const T* X = (const T*)a.X;
        const TB* S = (const TB*)a.Sop;
        const bool s_kfast = (a.ss_k == 1);
        const int mx = FM >= 0 ? (FM & 3) : stage_mode<T>(X, a.ldx, 1, a.n, 0, a.k1);
        const int ms = FM >= 0 ? ((FM >> 2) & 3) : stage_mode<TB>(S, a.ss_n, a.ss_k, a.c, 0, a.k1);
        // register prefetch of the next K tile while the current one feeds the matrix cores
        if (a.k1 > 0) {
            stage_load<T, BM, BK>(ra, X, a.ldx, 1, bm0, 0, a.n, a.k1, tid, mx);
            stage_load<TB, BN, BK>(rb, S, a.ss_n, a.ss_k, bn0, 0, a.c, a.k1, tid, ms);
        }
        for (int k0 = 0; k0 < a.k1; k0 += BK) {
            __syncthreads();
            if (k0 + BK <= a.k1) {             // (a whole K tile: no tail masks -- uniform)
                stage_store<T, T, BM, BK, LDA, true>(As, ra, true, k0, a.k1, AOP_NONE, tid, mx);
                stage_store<T, TB, BN, BK, LDB, true>(Bs, rb, s_kfast, k0, a.k1, AOP_NONE, tid, ms);
            } else {
                stage_store<T, T, BM, BK, LDA>(As, ra, true, k0, a.k1, AOP_NONE, tid, mx);
                stage_store<T, TB, BN, BK, LDB>(Bs, rb, s_kfast, k0, a.k1, AOP_NONE, tid, ms);
            }
            __syncthreads();
            if (k0 + BK < a.k1) {
                stage_load<T, BM, BK>(ra, X, a.ldx, 1, bm0, k0 + BK, a.n, a.k1, tid, mx);
                stage_load<TB, BN, BK>(rb, S, a.ss_n, a.ss_k, bn0, k0 + BK, a.c, a.k1, tid, ms);
            }
#pragma unroll
            for (int kk = 0; kk < BK; kk += MF::KT) {
                T av[WR], bv[WC];
                const int kr = kk + MF::ab_k(lane);
#pragma unroll
                for (int i = 0; i < WR; ++i) av[i] = As[kr][wm0 + i * MF::MT + MF::a_row(lane)];
#pragma unroll
                for (int j = 0; j < WC; ++j) bv[j] = Bs[kr][wn0 + j * MF::NT + MF::a_row(lane)];
#pragma unroll
                for (int i = 0; i < WR; ++i)
#pragma unroll
                    for (int j = 0; j < WC; ++j) accE[i][j] = MF::mma(av[i], bv[j], accE[i][j]);
            }
        }
    }
#ifdef SKF_PROBE_STAMPS
    const long long su1 = wall_clock64();
#endif
    // ---- split A into (A+, A-)
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                T v = accE[i][j][r];
                if (a.nan_to_num) v = nan_to_num(v);
                accE[i][j][r] = v > (T)0 ? v : (T)0;
                accD[i][j][r] = v > (T)0 ? (T)0 : -v;
            }
    // ---- phase 2: accE += G * Bn ; accD += G * Bp   (one staging of the G tile feeds both)
    if (a.phase2) {
        const T* G = (const T*)a.G;
        const TB* Bn = (const TB*)a.Bn;
        const TB* Bp = (const TB*)a.Bp;
        const int mg = FM >= 0 ? ((FM >> 4) & 3) : stage_mode<T>(G, a.ldg, 1, a.n, 0, a.c);
        const int mbn = FM >= 0 ? ((FM >> 6) & 3) : stage_mode<TB>(Bn, 1, a.ldb, a.c, 0, a.c);
        const int mbp = FM >= 0 ? ((FM >> 6) & 3) : stage_mode<TB>(Bp, 1, a.ldb, a.c, 0, a.c);
        stage_load<T, BM, BK>(ra, G, a.ldg, 1, bm0, 0, a.n, a.c, tid, mg);
        stage_load<TB, BN, BK>(rb, Bn, 1, a.ldb, bn0, 0, a.c, a.c, tid, mbn);
        stage_load<TB, BN, BK>(rb2, Bp, 1, a.ldb, bn0, 0, a.c, a.c, tid, mbp);
#ifdef SKF_PROBE_STAMPS
        long long pw[4] = {0, 0, 0, 0}, pt = wall_clock64();
#define SKF_PSTAMP(i) { const long long now_ = wall_clock64(); pw[i] += now_ - pt; pt = now_; }
#else
#define SKF_PSTAMP(i)
#endif
        for (int k0 = 0; k0 < a.c; k0 += BK) {
            __syncthreads();
            SKF_PSTAMP(0)
            if (k0 + BK <= a.c) {
                stage_store<T, T, BM, BK, LDA, true>(As, ra, true, k0, a.c, AOP_NONE, tid, mg);
                stage_store<T, TB, BN, BK, LDB, true>(Bs, rb, a.ldb == 1, k0, a.c, AOP_NONE, tid, mbn);
                stage_store<T, TB, BN, BK, LDB, true>(Bs2, rb2, a.ldb == 1, k0, a.c, AOP_NONE, tid, mbp);
            } else {
                stage_store<T, T, BM, BK, LDA>(As, ra, true, k0, a.c, AOP_NONE, tid, mg);
                stage_store<T, TB, BN, BK, LDB>(Bs, rb, a.ldb == 1, k0, a.c, AOP_NONE, tid, mbn);
                stage_store<T, TB, BN, BK, LDB>(Bs2, rb2, a.ldb == 1, k0, a.c, AOP_NONE, tid, mbp);
            }
            SKF_PSTAMP(1)
            __syncthreads();
            SKF_PSTAMP(2)
            if (k0 + BK < a.c) {
                stage_load<T, BM, BK>(ra, G, a.ldg, 1, bm0, k0 + BK, a.n, a.c, tid, mg);
                stage_load<TB, BN, BK>(rb, Bn, 1, a.ldb, bn0, k0 + BK, a.c, a.c, tid, mbn);
                stage_load<TB, BN, BK>(rb2, Bp, 1, a.ldb, bn0, k0 + BK, a.c, a.c, tid, mbp);
            }
#pragma unroll
            for (int kk = 0; kk < BK; kk += MF::KT) {
                T av[WR], bv[WC], bv2[WC];
                const int kr = kk + MF::ab_k(lane);
#pragma unroll
                for (int i = 0; i < WR; ++i) av[i] = As[kr][wm0 + i * MF::MT + MF::a_row(lane)];
#pragma unroll
                for (int j = 0; j < WC; ++j) {
                    bv[j] = Bs[kr][wn0 + j * MF::NT + MF::a_row(lane)];
                    bv2[j] = Bs2[kr][wn0 + j * MF::NT + MF::a_row(lane)];
                }
#pragma unroll
                for (int i = 0; i < WR; ++i)
#pragma unroll
                    for (int j = 0; j < WC; ++j) {
                        accE[i][j] = MF::mma(av[i], bv[j], accE[i][j]);
                        accD[i][j] = MF::mma(av[i], bv2[j], accD[i][j]);
                    }
            }
            SKF_PSTAMP(3)
        }
#ifdef SKF_PROBE_STAMPS
        if (tid == 0 && blockIdx.y == gridDim.y / 2 && blockIdx.x == 0 && a.n >= 20000)
            printf("side_update phase 2 loop n %d c %d: first barrier %lld, registers -> LDS (waits for the loads) %lld, second barrier %lld, issue + matrix cores %lld (x10 ns)\n",
                   a.n, a.c, pw[0], pw[1], pw[2], pw[3]);
#endif
    }
#ifdef SKF_PROBE_STAMPS
    const long long su2 = wall_clock64();
#endif
    // ---- epilogue
    T* E = (T*)a.E;
    T* D = (T*)a.D;
    // (accumulate: the old values of a tile are all loaded before the first store -- E and D may alias for all the compiler
    //  knows, and `E[idx] += x; D[idx] += y` element by element was one memory round trip per element: 14-29 us of the 25-37 a
    //  workgroup of config 5's movie side took, time stamps of a probe build)
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WC; ++j) {
            T eo[MF::NREG], dq[MF::NREG];
            const int n = bn0 + wn0 + j * MF::NT + MF::d_col(lane);
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int m = bm0 + wm0 + i * MF::MT + MF::d_row(lane, r);
                const int64_t idx = (int64_t)m * a.lde + n;
                const bool in = m < a.n && n < a.c;
                eo[r] = (a.accumulate && in) ? E[idx] : (T)0;
                dq[r] = (a.accumulate && in) ? D[idx] : (T)0;
            }
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int m = bm0 + wm0 + i * MF::MT + MF::d_row(lane, r);
                if (m < a.n && n < a.c) {
                    const int64_t idx = (int64_t)m * a.lde + n;
                    if (a.accumulate) {
                        E[idx] = eo[r] + accE[i][j][r];
                        D[idx] = dq[r] + accD[i][j][r];
                    } else {
                        E[idx] = accE[i][j][r];
                        D[idx] = accD[i][j][r];
                    }
                }
            }
        }
#ifdef SKF_PROBE_STAMPS
    if (tid == 0 && (blockIdx.y == 0 || blockIdx.y == gridDim.y / 2) && blockIdx.x == 0 && a.n >= 20000)
        printf("side_update n %d c %d k1 %d phase2 %d acc %d block %d: phase 1 %lld, split + phase 2 %lld, epilogue %lld (x10 ns), start %lld\n", a.n, a.c, a.k1,
               a.phase2, a.accumulate, (int)blockIdx.y, su1 - su0, su2 - su1, wall_clock64() - su2, su0);
#endif
}

// ------------------------------------------------------------------------------------------
// One fold-in iteration (reference _dfmf.py:385-428, the target type without constraints) in ONE launch, for up to
// FOLD_MAXB plans of one graph at once (blockIdx.z = the plan: the restarts the reference hands to joblib workers,
// dfmf.py:191-199).  Everything that does not depend on the folded-in factor was summed once (Ec, Dc, Bn = sum B-,
// Bp = sum B+: prepare_transform), so an iteration is
//     E = Ec + G * Bn ;  D = Dc + G * Bp ;  Gout = G * sqrt(E / max(D, eps))
// -- two contractions that share their A tile, the update in the epilogue, E and D never written.  G and Gout are two
// buffers (another column tile of the same rows still reads the old rows); the host swaps them after every launch.
// G, Ec, Dc, Gout are T (the master type), Bn / Bp are f64 c x c matrices rounded to T while staged.
// ------------------------------------------------------------------------------------------
#define FOLD_MAXB 16
struct FoldArgs {
    const void* G[FOLD_MAXB];
    void* Gout[FOLD_MAXB];
    const void* Bn[FOLD_MAXB];
    const void* Bp[FOLD_MAXB];
    const void* Ec[FOLD_MAXB];
    const void* Dc[FOLD_MAXB];
    int n, c;
};

template <typename T, int WR, int WC, int BK>
__global__ __launch_bounds__(GEMM_THREADS, (sizeof(T) == 4 ? 2 : 1)) void foldin_step_kernel(FoldArgs a) {
    typedef Mfma<T> MF;
    constexpr int BM = 2 * WR * MF::MT, BN = 2 * WC * MF::NT;
    constexpr int LDA = BM + 1, LDB = BN + 1;
    __shared__ T As[BK][LDA];
    __shared__ T Bs[BK][LDB];
    __shared__ T Bs2[BK][LDB];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (WR * MF::MT), wn0 = (wave & 1) * (WC * MF::NT);
    const int bm0 = blockIdx.y * BM, bn0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const T* __restrict__ G = (const T*)a.G[z];
    const double* __restrict__ Bn = (const double*)a.Bn[z];
    const double* __restrict__ Bp = (const double*)a.Bp[z];
    const int64_t ld = a.c;

    typename MF::acc_t accE[WR][WC], accD[WR][WC];
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) accE[i][j][r] = accD[i][j][r] = (T)0;

    T ra[BM * BK / GEMM_THREADS];
    double rb[BN * BK / GEMM_THREADS], rb2[BN * BK / GEMM_THREADS];
    const int mg = stage_mode<T>(G, ld, 1, a.n, 0, a.c);
    const int mbn = stage_mode<double>(Bn, 1, ld, a.c, 0, a.c);
    const int mbp = stage_mode<double>(Bp, 1, ld, a.c, 0, a.c);
    stage_load<T, BM, BK>(ra, G, ld, 1, bm0, 0, a.n, a.c, tid, mg);
    stage_load<double, BN, BK>(rb, Bn, 1, ld, bn0, 0, a.c, a.c, tid, mbn);
    stage_load<double, BN, BK>(rb2, Bp, 1, ld, bn0, 0, a.c, a.c, tid, mbp);
    for (int k0 = 0; k0 < a.c; k0 += BK) {
        __syncthreads();
        stage_store<T, T, BM, BK, LDA>(As, ra, true, k0, a.c, AOP_NONE, tid, mg);
        stage_store<T, double, BN, BK, LDB>(Bs, rb, ld == 1, k0, a.c, AOP_NONE, tid, mbn);
        stage_store<T, double, BN, BK, LDB>(Bs2, rb2, ld == 1, k0, a.c, AOP_NONE, tid, mbp);
        __syncthreads();
        if (k0 + BK < a.c) {
            stage_load<T, BM, BK>(ra, G, ld, 1, bm0, k0 + BK, a.n, a.c, tid, mg);
            stage_load<double, BN, BK>(rb, Bn, 1, ld, bn0, k0 + BK, a.c, a.c, tid, mbn);
            stage_load<double, BN, BK>(rb2, Bp, 1, ld, bn0, k0 + BK, a.c, a.c, tid, mbp);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += MF::KT) {
            T av[WR], bv[WC], bv2[WC];
            const int kr = kk + MF::ab_k(lane);
#pragma unroll
            for (int i = 0; i < WR; ++i) av[i] = As[kr][wm0 + i * MF::MT + MF::a_row(lane)];
#pragma unroll
            for (int j = 0; j < WC; ++j) {
                bv[j] = Bs[kr][wn0 + j * MF::NT + MF::a_row(lane)];
                bv2[j] = Bs2[kr][wn0 + j * MF::NT + MF::a_row(lane)];
            }
#pragma unroll
            for (int i = 0; i < WR; ++i)
#pragma unroll
                for (int j = 0; j < WC; ++j) {
                    accE[i][j] = MF::mma(av[i], bv[j], accE[i][j]);
                    accD[i][j] = MF::mma(av[i], bv2[j], accD[i][j]);
                }
        }
    }
    // ---- epilogue: the multiplicative update (_dfmf.py:427-428)
    const T eps = (T)2.220446049250313e-16;
    const T* __restrict__ Ec = (const T*)a.Ec[z];
    const T* __restrict__ Dc = (const T*)a.Dc[z];
    T* __restrict__ Gout = (T*)a.Gout[z];
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int m = bm0 + wm0 + i * MF::MT + MF::d_row(lane, r);
                const int n = bn0 + wn0 + j * MF::NT + MF::d_col(lane);
                if (m < a.n && n < a.c) {
                    const int64_t idx = (int64_t)m * ld + n;
                    const T e = Ec[idx] + accE[i][j][r];
                    const T d = Dc[idx] + accD[i][j][r];
                    const T den = (d > eps || d != d) ? d : eps;         // np.maximum(D, eps) (NaN propagates)
                    Gout[idx] = G[idx] * (T)sqrt(e / den);
                }
            }
}

// ------------------------------------------------------------------------------------------
// vector-ALU GEMM with the same contract (64 x 64 block tile, 4 x 4 outputs per thread).
// ------------------------------------------------------------------------------------------
template <typename T, typename TA, typename TB>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_valu_kernel(GemmArgs g) {
    constexpr int BM = 64, BN = 64, BK = 16, LDT = BM + 1;
    __shared__ T As[BK][LDT];
    __shared__ T Bs[BK][LDT];
    if (g.gate != nullptr && *g.gate == 0) return;                         // (uniform)
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;          // 16 x 16 threads, 4 x 4 outputs each
    const int bm0 = blockIdx.y * BM, bn0 = blockIdx.x * BN;
    const int kz0 = blockIdx.z * g.k_chunk;
    const int kz1 = (kz0 + g.k_chunk < g.K) ? kz0 + g.k_chunk : g.K;
    const TA* __restrict__ A = (const TA*)g.A;
    const TB* __restrict__ B = (const TB*)g.B;
    const bool a_kfast = (g.sa_k == 1), b_kfast = (g.sb_k == 1);
    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (T)0;
    TA ra[BM * BK / GEMM_THREADS];
    TB rb[BN * BK / GEMM_THREADS];
    for (int k0 = kz0; k0 < kz1; k0 += BK) {
        stage_load<TA, BM, BK>(ra, A, g.sa_m, g.sa_k, bm0, k0, g.M, kz1, tid);
        stage_load<TB, BN, BK>(rb, B, g.sb_n, g.sb_k, bn0, k0, g.N, kz1, tid);
        __syncthreads();
        stage_store<T, TA, BM, BK, LDT>(As, ra, a_kfast, k0, kz1, g.aop, tid);
        stage_store<T, TB, BN, BK, LDT>(Bs, rb, b_kfast, k0, kz1, AOP_NONE, tid);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
    }
    const bool split = (gridDim.z > 1);
    T sq = (T)0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = bm0 + ty + 16 * i, n = bn0 + tx + 16 * j;
            if (m < g.M && n < g.N) {
                const T v = acc[i][j];
                if (split) {
                    ((T*)g.part)[((int64_t)blockIdx.z * g.M + m) * g.N + n] = v;
                } else if (g.epi == EPI_SQDIFF) {
                    const int64_t ci = (int64_t)m * g.ldc + n;
                    const T rv = g.c_bf16 ? (T)bf16_to_f32(((const uint16_t*)g.C)[ci]) : ((const T*)g.C)[ci];
                    const T d = rv - v;
                    sq += d * d;
                } else {
                    epilogue_store<T>(g, m, n, v);
                }
            }
        }
    if (!split && g.epi == EPI_SQDIFF) {
        __shared__ T red[GEMM_THREADS / 64];
        sq = wave_sum(sq);
        if ((tid & 63) == 0) red[tid >> 6] = sq;
        __syncthreads();
        if (tid == 0) {
            T s = (T)0;
            for (int w = 0; w < GEMM_THREADS / 64; ++w) s += red[w];
            ((T*)g.C2)[blockIdx.y * gridDim.x + blockIdx.x] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 relation contraction (the MFMA-roofline kernel of the SKF_BF16 engine):
//     C[M x N] (f32) = A[M x Kp] (bf16, row-major) * Bt[N x Kp]^T (bf16, row-major)
// i.e. both operands K-contiguous: A = R (for P = R G_j) or the stored transpose R^T (for
// Q = R^T G_i), Bt = the bf16 transpose G^T of a factor.  Kp (the padded inner dimension) is a
// multiple of 64 and the padding is zero-filled by the engine, so the K loop has no tail.
//
// Workgroup: 256 threads = 4 waves (2 x 2); block tile 128 x BN (BN = 128 or 256: the whole
// factor rank in one tile, so R streams from HBM exactly once), K tile 64.
// v_mfma_f32_16x16x32_bf16: lane l holds A[row = l&15][k = 8*(l>>4) .. +7] -- eight consecutive
// K elements = one 16-byte LDS read.  LDS tiles are [rows][64] bf16 (128-byte rows, no padding)
// with the 16-byte chunk index XOR-swizzled by (row & 7): conflict-free ds_read_b128 for the
// fragment pattern and conflict-free ds_write_b128 for the staging pattern (8 consecutive
// lanes fill one row).  Global loads are 16 B per lane, 128 B contiguous per row.
// One K step: [global loads of tile t+1 in flight] 2 x {fragment reads, MFMAs} | barrier |
//             registers -> LDS | barrier.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Bf16GemmArgs {
    const uint16_t* A;     // [M][lda]
    const uint16_t* Bt;    // [N][ldb]
    float* C;              // [M][ldc]           (gridDim.z == 1)
    float* part;           // [gridDim.z][M][N]  (split-K partials otherwise)
    int64_t lda, ldb, ldc;
    int M, N, Kp;          // Kp % 64 == 0
    int k_chunk;           // K range per z-slice, multiple of 64
    int64_t a_kstep, b_kstep;   // elements between consecutive 64-wide K tiles of a row: 64 for the
                                // row-major layout; rows*64 for the K-tile-major layout, in which the
                                // [rows][64] slab of one K tile is contiguous (lda = ldb = 64)
    // elementwise epilogues against a stored bf16 relation (gemm_bf16_v2_kernel<.., EPI_T_*>: the product is
    // taken transposed there -- M = relation columns, N = relation rows)
    uint16_t* R;                // [N][ldr] bf16, ldr % 8 == 0
    int64_t ldr;
    const uint8_t* mbits;       // EPI_T_COMPLETE: packed mask [N][ldmb bytes], ldmb % 16 == 0, bits >= M are zero
    int64_t ldmb;
    double* sq;                 // EPI_T_SQERR: one partial per workgroup (blockIdx.y * gridDim.x + blockIdx.x)
    // EPI_T_COMPLETE with the known entries of every 256 x 256 tile as a compact list (known_fill_kernel):
    // entries koff[t] .. koff[t+1]-1 of klist belong to tile t = blockIdx.y * gridDim.x + blockIdx.x, each
    // (m_loc * 256 + n_loc) | bf16 value << 16.  klist == nullptr: blend through the mask instead.
    const uint32_t* koff;
    const uint32_t* klist;
    // EPI_T_SQERR of a binary relation stored as a bitmap (ABITS): bit (n & 7) of Rbits[m * ldrbits + (n >> 3)] = R[m][n]
    const uint8_t* Rbits;
    int64_t ldrbits;
};
// epilogue of gemm_bf16_v2_kernel: store the f32 tile | DFMC completion | squared residual
enum { EPI_T_STORE = 0, EPI_T_COMPLETE = 1, EPI_T_SQERR = 2 };

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return row * 8 + (chunk ^ (row & 7)); }

constexpr int KPF = 3;       // known entries per thread that EPI_T_COMPLETE fetches before the K loop of its tile
template <int NTHR>
__device__ __forceinline__ void bf16_tile_prefetch_known(const Bf16GemmArgs& g, int tile, uint32_t& kq0, uint32_t& kq1,
                                                         uint32_t (&kpre)[KPF]);
template <int EPI, int BM, int BN, int NTHR, int NJ>
__device__ __forceinline__ void bf16_tile_epilogue(const Bf16GemmArgs& g, f32x4 (&acc)[4][NJ], u32x4* smem, int bm0, int bn0,
                                                   int tile, uint32_t kq0, uint32_t kq1, const uint32_t (&kpre)[KPF]);

// 128 x BN tile, 4 waves, register-staged double buffering.  EPI_T_STORE: the P-type products of small problems.
// EPI_T_COMPLETE / EPI_T_SQERR: the elementwise passes over a stored relation (see bf16_tile_epilogue) -- their K loop is
// a handful of tiles and the phases of a tile (operand latency, MFMAs, staging, write-out) are serial inside a workgroup;
// with 70 KiB of LDS and 256 threads TWO workgroups share a CU and one's write-out runs under the other's K loop.
template <int BN, int TAG, int EPI = EPI_T_STORE>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(Bf16GemmArgs g) {
    constexpr int BM = 128, BK = 64;
    constexpr int WN = BN / 2;              // wave tile 64 x WN
    constexpr int NJ = WN / 16;             // 16-wide column blocks per wave
    constexpr int A_PER = BM * 8 / 256;     // 16-byte chunks per thread and tile
    constexpr int B_PER = BN * 8 / 256;
    HIP_DYNAMIC_SHARED(u32x4, smem)          // (BM + BN) * 128 bytes of operand tiles; EPI_T_*: >= BN * (BM + 8) * 2 for the staged tile
    u32x4* As = smem;
    u32x4* Bs = smem + BM * 8;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * WN;
    const int bm0 = blockIdx.y * BM, bn0 = blockIdx.x * BN;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;     // EPI_T_*: index of the tile in koff / sq
    const int kz0 = blockIdx.z * g.k_chunk;
    const int kz1 = (kz0 + g.k_chunk < g.Kp) ? kz0 + g.k_chunk : g.Kp;
    const int nkt = (kz1 - kz0) / BK;
    const int srow = tid >> 3, schunk = tid & 7;        // staging: 8 lanes per 128-byte row
    uint32_t kq0 = 0u, kq1 = 0u, kpre[KPF] = {0u, 0u, 0u};
    if constexpr (EPI == EPI_T_COMPLETE) bf16_tile_prefetch_known<256>(g, tile, kq0, kq1, kpre);

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[A_PER], rb[B_PER];
    const u32x4 zero = {0u, 0u, 0u, 0u};

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            const int m = bm0 + srow + 32 * p;
            ra[p] = (m < g.M) ? *(const u32x4*)(g.A + (int64_t)m * g.lda + (int64_t)(k0 >> 6) * g.a_kstep + schunk * 8) : zero;
        }
#pragma unroll
        for (int p = 0; p < B_PER; ++p) {
            const int n = bn0 + srow + 32 * p;
            rb[p] = (n < g.N) ? *(const u32x4*)(g.Bt + (int64_t)n * g.ldb + (int64_t)(k0 >> 6) * g.b_kstep + schunk * 8) : zero;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int p = 0; p < A_PER; ++p) As[swz_chunk(srow + 32 * p, schunk)] = ra[p];
#pragma unroll
        for (int p = 0; p < B_PER; ++p) Bs[swz_chunk(srow + 32 * p, schunk)] = rb[p];
    };

    if (nkt > 0) {
        load_tiles(kz0);
        store_tiles();
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1 < nkt);
        if (more) load_tiles(kz0 + (kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = 4 * ks + (lane >> 4);
            bf16x8 a[4], b[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = __builtin_bit_cast(bf16x8, As[swz_chunk(wm0 + i * 16 + (lane & 15), chunk)]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = __builtin_bit_cast(bf16x8, Bs[swz_chunk(wn0 + j * 16 + (lane & 15), chunk)]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) store_tiles();
        __syncthreads();
    }

    if constexpr (EPI != EPI_T_STORE) {
        // (the loop's last __syncthreads() has released the operand tiles: the staged tile may overwrite them)
        bf16_tile_epilogue<EPI, BM, BN, 256, NJ>(g, acc, smem, bm0, bn0, tile, kq0, kq1, kpre);
        return;
    }
    // epilogue: D reg r of a 16 x 16 tile -> row = 4*(lane>>4) + r, col = lane & 15
    float* out = (gridDim.z > 1) ? g.part + (int64_t)blockIdx.z * g.M * g.N : g.C;
    const int64_t ldo = (gridDim.z > 1) ? g.N : g.ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = bm0 + wm0 + i * 16 + 4 * (lane >> 4) + r;
                const int n = bn0 + wn0 + j * 16 + (lane & 15);
                if (m < g.M && n < g.N) out[(int64_t)m * ldo + n] = acc[i][j][r];
            }
}

// ------------------------------------------------------------------------------------------
// bf16 relation contraction, the kernel of the two products that stream a relation:
//     AT = false:  C[M x N] (f32) = A[M x Kp] * Bt[N x Kp]^T      P = R G_j     (A = R,   K-contiguous rows)
//     AT = true :  C[M x N] (f32) = A[Kp x M]^T * Bt[N x Kp]^T    Q = R^T G_i   (A = the SAME row-major R:
//                  its rows are the contraction index, its columns the output rows -- no stored transpose)
// 256 x BN block tile, 512 threads = 8 waves (4 x 2, wave tile 64 x BN/2), K tile 64,
// v_mfma_f32_16x16x32_bf16 (lane l holds A[row = l&15][k = 8*(l>>4) .. +7]).
//
// Tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: one wave instruction moves 1 KiB into a
// wave-LINEAR LDS range, so the bank swizzle is applied to the per-lane SOURCE address); no staging
// registers, no ds_write pass.  The A operand (the relation, streamed from HBM, nt policy) lives in a ring
// of three 32 KiB buffers, the B operand (G^T, served by L2) in three buffers at BN = 128 and two at BN = 256
// (3 x 32 + 2 x 32 KiB = the whole LDS).  The DMA instructions are issued one at a time in slots BETWEEN the
// wave's groups of MFMAs, never in a burst behind a barrier (+3 %, profiles/r02_contraction_bounds.txt).
//
// K loop (round 2, second half): a K tile is consumed in two phases of 32 k.  The fragments of the NEXT phase are
// read while the matrix cores work through the current one -- into the registers the MFMAs have just released,
// plus one spare B fragment -- by inline-assembly ds_read_b128 / ds_read_b64_tr_b16 whose lgkmcnt the loop counts
// itself (the compiler's waitcnt pass knows nothing of them, so no s_waitcnt it would place can serialise the
// LDS-DMA ring).  The workgroup meets at ONE raw s_barrier per K tile, between the phases, behind a counted
// s_waitcnt vmcnt(PWA) that leaves exactly the newest A tile in flight (__syncthreads() would drain every
// LDS-DMA).  Before: 12 fragment reads, lgkmcnt(0), 32 MFMAs, twice per K tile, all 8 waves bursting at the LDS
// together: MFMA busy 57 -> 61 %, P12 2.56 -> 2.39 ms, compute-only probe 1.85 -> 1.57 ms.
// ABITS (bitmap operand): the bits of a tile are fetched by an inline-assembly global load two tiles ahead and
// expanded to bf16 0 / 1 into the A ring by inline-assembly ds_write_b128 in the same slots; every vmcnt / lgkmcnt
// of that path is counted by hand for the same reason.
//
// LDS images
//   AT = false: A tile [256 rows][64 k] and B tile [BN rows][64 k], 128-byte rows, 16-byte chunk index
//               XOR-swizzled by (row & 7): conflict-free ds_read_b128 fragment reads.
//   AT = true : A tile [64 k][256 m], 512-byte rows (one wave instruction = two k rows); a fragment is two
//               ds_read_b64_tr_b16 (lane i of a 16-lane group, element j <- the halfword that lane
//               4j + (i>>2) addresses, its element i&3: each group reads a [4 k][16 m] block and every lane
//               leaves with the 4 k values of ITS m).  Chunk pairs (32 bytes = the 16 m of one group) are
//               XOR-swizzled by ((k & 3) | ((k >> 3) & 1) << 2): the 8 k rows a half-wave touches fall
//               into 8 different 32-byte bank slots.
// Kp % 64 == 0 and the padding is zero-filled by the engine (AT: padding ROWS of A), so there is no K tail.
// Rows / columns past the end are clamped into the matrix (no divergent branches around the loads): they
// only feed accumulator rows / columns >= M / N, which are never stored.
// ------------------------------------------------------------------------------------------
// cache policy of the relation stream (aux operand of global_load_lds): 2 = nt, the tile is read once
// and never again by this or any other workgroup (+2 %; on the G^T stream nt measured -7 %).
#ifndef SKF_A_AUX
#define SKF_A_AUX 2
#endif
// (An alternative LDS image of the transposed-A tile -- [k/4][m/16] blocks of [4 k][16 m], every 16-lane group of a
// ds_read_b64_tr_b16 reading one contiguous 128-byte block -- measured equal within 1 %: the XOR-swizzled rows are
// not bank-conflict bound; profiles/r02_contraction_bounds.txt.)
#ifndef SKF_B_AUX
#define SKF_B_AUX 0
#endif
// What the K loop of gemm_bf16_v2_kernel does.  The product: everything.  (The bound-finding builds of round 2 --
// ingest-only, compute-only, one stream removed; profiles/r02_contraction_bounds.txt -- instantiate the kernel with a
// policy of their own from tools/probe/v2_policies.h; nothing of them is compiled here.)
struct V2Full {
    static constexpr bool stream_a = true;     // LDS-DMA of the relation tiles inside the loop
    static constexpr bool stream_b = true;     // LDS-DMA of the G^T tiles inside the loop
    static constexpr bool mfma = true;         // the matrix-core instructions
};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// swizzle key of a k row of the transposed A image (AT)
__device__ __forceinline__ int at_key(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// ABITS: the A operand is a BINARY relation stored as a bitmap (1 bit per entry instead of a bf16: 1/16 of the
// bytes from HBM): g.A points at the bitmap, g.lda is its row pitch in BYTES (a multiple of 8); a row of the bitmap
// is a row of the relation, bit (c & 7) of byte c >> 3.  Every thread fetches the 32 bits of its quarter row of the
// NEXT-but-two tile into a register, and expands them into bf16 0 / 1 in the LDS image of the tile (the same swizzled
// images as the LDS-DMA path, so fragment reads and MFMAs are unchanged) in the slots between the MFMA groups.
__device__ __forceinline__ u32x4 bits_to_bf16x8(uint32_t byte) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        v[j] = ((byte >> (2 * j)) & 1u) * 0x3F80u | ((byte >> (2 * j + 1)) & 1u) * 0x3F800000u;
    return v;
}

template <int NTHR>
__device__ __forceinline__ void bf16_tile_prefetch_known(const Bf16GemmArgs& g, int tile, uint32_t& kq0, uint32_t& kq1,
                                                         uint32_t (&kpre)[KPF]) {
    kq0 = kq1 = 0u;
#pragma unroll
    for (int q = 0; q < KPF; ++q) kpre[q] = 0u;
    if (g.klist != nullptr) {
        kq0 = g.koff[tile];
        kq1 = g.koff[tile + 1];
#pragma unroll
        for (int q = 0; q < KPF; ++q) {
            const uint32_t k = kq0 + threadIdx.x + q * NTHR;
            if (k < kq1) kpre[q] = __builtin_nontemporal_load(g.klist + k);
        }
    }
}

// Elementwise epilogues of the bf16 tile kernels (gemm_bf16_v2_kernel: BM = 256, 512 threads; gemm_bf16_kernel:
// BM = 128, 256 threads -- two workgroups per CU, so that one tile's write-out runs under the other's K loop).
// BM = tile rows = relation COLUMNS, BN = tile columns = relation ROWS; waves as (wave >> 1) x (wave & 1) of 64 x BN/2.
template <int EPI, int BM, int BN, int NTHR, int NJ>
__device__ __forceinline__ void bf16_tile_epilogue(const Bf16GemmArgs& g, f32x4 (&acc)[4][NJ], u32x4* smem, int bm0, int bn0,
                                                   int tile, uint32_t kq0, uint32_t kq1, const uint32_t (&kpre)[KPF]) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * (BN / 2);
    // The passes that touch a relation ELEMENTWISE against its reconstruction T = H G_j^T (K = c_j: a handful
    // of K tiles -- the pass is bound by the relation's bytes).  The product is taken TRANSPOSED: the kernel's
    // A operand is bf16(G_j) (tile rows = relation COLUMNS n), its Bt operand bf16(H) (tile columns = relation
    // ROWS m), so that the four accumulator elements of a lane are four CONSECUTIVE columns of one relation row:
    //   EPI_T_COMPLETE  DFMC completion (_dfmc.py:319-325): R[m][n] = bf16(T[m][n]) where the mask bit is set;
    //                   the tile is staged as bf16 in LDS (one 8-byte store per 16 x 16 block and lane) and written
    //                   as whole 16-byte chunks along the rows of R, the old chunk being read only when it holds
    //                   a known entry to keep
    //   EPI_T_SQERR     partial of sum (R - T)^2: the R tile is staged in LDS with coalesced 16-byte loads and
    //                   every lane compares its accumulator elements with it (f32 product, stored bf16 entries)
    // (the last barrier of the K loop has released the operand rings: the whole LDS is free)
    constexpr int TLD = BM + 8;              // halfwords per staged relation row: 528 B (16-byte aligned rows)
    constexpr int CH = BM / 8;               // 16-byte chunks per staged row
    uint16_t* T = (uint16_t*)smem;           // T[m_loc][n_loc], m_loc < BN (relation rows), n_loc < BM (relation columns)
    const int rel_rows = g.N, rel_cols = g.M, row0 = bn0, col0 = bm0;
    if constexpr (EPI == EPI_T_COMPLETE) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nl = wm0 + i * 16 + 4 * (lane >> 4), ml = wn0 + j * 16 + (lane & 15);
                const uint32_t lo = (uint32_t)f32_to_bf16_rne(acc[i][j][0]) | ((uint32_t)f32_to_bf16_rne(acc[i][j][1]) << 16);
                const uint32_t hi = (uint32_t)f32_to_bf16_rne(acc[i][j][2]) | ((uint32_t)f32_to_bf16_rne(acc[i][j][3]) << 16);
                uint32_t* dst = (uint32_t*)(T + ml * TLD + nl);
                dst[0] = lo; dst[1] = hi;
            }
        __syncthreads();
        constexpr int NIT = BN * CH / NTHR;
        if (g.klist != nullptr) {
            // Known entries of this tile from their compact list straight into the staged tile, then EVERY chunk
            // is written out: full-line streaming stores, no read of the old relation, no mask traffic
            // (at 2 % known entries the blend below reads 3 of 4 lines of R back).
#pragma unroll
            for (int q = 0; q < KPF; ++q)
                if (kq0 + tid + q * NTHR < kq1) T[((kpre[q] >> 8) & 0xFFu) * TLD + (kpre[q] & 0xFFu)] = (uint16_t)(kpre[q] >> 16);
            for (uint32_t k = kq0 + tid + KPF * NTHR; k < kq1; k += NTHR) {
                const uint32_t e = g.klist[k];
                T[((e >> 8) & 0xFFu) * TLD + (e & 0xFFu)] = (uint16_t)(e >> 16);
            }
            __syncthreads();
            constexpr int NB = 8;            // chunks out of LDS per batch (one wait), then their stores
            static_assert(NIT % NB == 0, "write-out batches");
#pragma unroll
            for (int q0 = 0; q0 < NIT; q0 += NB) {
            u32x4 tv[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int it = tid + (q0 + q) * NTHR;
                tv[q] = *(const u32x4*)(T + (it / CH) * TLD + (it % CH) * 8);
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int it = tid + (q0 + q) * NTHR;
                const int r = it / CH, c = it % CH;
                const int left = rel_cols - (col0 + c * 8);           // columns of this chunk inside the relation
                if (row0 + r >= rel_rows || left <= 0) continue;
                u32x4 v = tv[q];
                if (left < 8) {                                       // padding columns of R stay zero
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= left) v[e >> 1] &= ~(0xFFFFu << (16 * (e & 1)));
                }
                *(u32x4*)(g.R + (int64_t)(row0 + r) * g.ldr + col0 + c * 8) = v;
            }
            }
        } else {
        // item = 8 consecutive columns of one relation row; BM / 8 lanes cover the contiguous bytes of a tile row.
        // All mask bytes of the thread first (independent loads, one wait), then the stores.
        uint32_t mb[NIT];
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int it = tid + q * NTHR;
            const int r = it / CH, c = it % CH;
            const int m = row0 + r;
            const bool in = m < rel_rows && (int64_t)(col0 >> 3) + c < g.ldmb;
            mb[q] = in ? g.mbits[(int64_t)m * g.ldmb + (col0 >> 3) + c] : 0u;   // bit b: column col0 + 8c + b is unknown
        }
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            if (mb[q] == 0u) continue;
            const int it = tid + q * NTHR;
            const int r = it / CH, c = it % CH;
            u32x4 v = *(const u32x4*)(T + r * TLD + c * 8);
            u32x4* dst = (u32x4*)(g.R + (int64_t)(row0 + r) * g.ldr + col0 + c * 8);
            if (mb[q] != 0xFFu) {
                const u32x4 old = *dst;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!(mb[q] & (1u << e))) {
                        const uint32_t sh = 16 * (e & 1), msk = 0xFFFFu << sh;
                        v[e >> 1] = (v[e >> 1] & ~msk) | (old[e >> 1] & msk);
                    }
            }
            *dst = v;
        }
        }
    } else {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int it = tid; it < BN * CH; it += NTHR) {       // zero outside the stored matrix: never compared
            const int r = it / CH, c = it % CH;
            const int m = row0 + r;
            const int64_t col = (int64_t)col0 + c * 8;
            u32x4 v = zero;
            if (g.Rbits != nullptr) {
                if (m < rel_rows && (col >> 3) < g.ldrbits) v = bits_to_bf16x8(g.Rbits[(int64_t)m * g.ldrbits + (col >> 3)]);
            } else if (m < rel_rows && col + 8 <= g.ldr) {
                v = *(const u32x4*)(g.R + (int64_t)m * g.ldr + col);
            }
            *(u32x4*)(T + r * TLD + c * 8) = v;
        }
        __syncthreads();
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nl = wm0 + i * 16 + 4 * (lane >> 4), ml = wn0 + j * 16 + (lane & 15);
                const uint32_t* src = (const uint32_t*)(T + ml * TLD + nl);
                const uint32_t lo = src[0], hi = src[1];
                const float rv[4] = {bf16_to_f32((uint16_t)(lo & 0xFFFFu)), bf16_to_f32((uint16_t)(lo >> 16)),
                                     bf16_to_f32((uint16_t)(hi & 0xFFFFu)), bf16_to_f32((uint16_t)(hi >> 16))};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + ml < rel_rows && col0 + nl + r < rel_cols) {
                        const float d = rv[r] - acc[i][j][r];
                        sacc += d * d;
                    }
            }
        const double ws = wave_sum((double)sacc);
        __syncthreads();                     // every wave is done with T: its first bytes carry the wave sums now
        double* wsum = (double*)smem;
        if (lane == 0) wsum[wave] = ws;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < NTHR / 64; ++w) tot += wsum[w];
            g.sq[tile] = tot;
        }
    }
}

template <int BN, int TAG, bool AT, int EPI = EPI_T_STORE, bool ABITS = false, class POL = V2Full>
__global__ __launch_bounds__(512) void gemm_bf16_v2_kernel(Bf16GemmArgs g) {
    static_assert(EPI == EPI_T_STORE || (BN == 256 && !AT), "the elementwise epilogues use the 256 x 256 P-form tile");
    constexpr int BM = 256, BK = 64;
    constexpr int WN = BN / 2;
    constexpr int NJ = WN / 16;             // 16-wide column blocks per wave
    constexpr int AST = 3;                                       // A ring depth
    constexpr int BST = (BN == 256) ? 2 : 3;                     // B ring depth
    constexpr int ASZ = BM * 8, BSZ = BN * 8;                   // u32x4 entries per buffer
    constexpr int PWA = BM / 64, PWB = BN / 64;                 // LDS-DMA instructions per wave and K tile
    HIP_DYNAMIC_SHARED(u32x4, smem)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * WN;
    const int kz0 = blockIdx.z * g.k_chunk;
    const int kz1 = (kz0 + g.k_chunk < g.Kp) ? kz0 + g.k_chunk : g.Kp;
    const int nkt = (kz1 - kz0) / BK;
    const int bm0 = blockIdx.y * BM, bn0 = blockIdx.x * BN;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;     // EPI_T_*: index of the tile in koff / sq

    // EPI_T_COMPLETE: the first known entries of the tile (3 per thread) travel while the K loop runs
    uint32_t kq0 = 0u, kq1 = 0u, kpre[KPF] = {0u, 0u, 0u};
    if constexpr (EPI == EPI_T_COMPLETE) bf16_tile_prefetch_known<512>(g, tile, kq0, kq1, kpre);

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // piece p of the direct global -> LDS copy of one K tile: wave w fills the 1 KiB blocks w*PW .. +PW-1
    const int rr = lane >> 3, pc = lane & 7;
    auto dma_A1 = [&](int k0, int buf, int p) {
        u32x4* Ad = smem + buf * ASZ;
        const int blk = wave * PWA + p;
        if constexpr (AT) {
            // block = k rows 2*blk, 2*blk+1 of the [64 k][256 m] image; lane -> (k row, physical chunk)
            const int kr = 2 * blk + (lane >> 5);
            const int c = (lane & 31) ^ (at_key(kr) << 1);             // logical 16-byte chunk (8 m)
            int64_t col = (int64_t)bm0 + c * 8;
            if (col > g.lda - 8) col = g.lda - 8;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(g.A + (int64_t)(k0 + kr) * g.lda + col),
                (__attribute__((address_space(3))) void*)(Ad + blk * 64), 16, 0, SKF_A_AUX);
        } else {
            const int row = blk * 8 + rr;
            const int c = pc ^ (row & 7);
            const int m = bm0 + row;
            const int mc = m < g.M ? m : g.M - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(g.A + (int64_t)mc * g.lda + (int64_t)(k0 >> 6) * g.a_kstep + c * 8),
                (__attribute__((address_space(3))) void*)(Ad + blk * 64), 16, 0, SKF_A_AUX);
        }
    };
    auto dma_B1 = [&](int k0, int buf, int p) {
        u32x4* Bd = smem + AST * ASZ + buf * BSZ;
        const int blk = wave * PWB + p;
        const int row = blk * 8 + rr;
        const int c = pc ^ (row & 7);
        const int n = bn0 + row;
        const int nc = n < g.N ? n : g.N - 1;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(g.Bt + (int64_t)nc * g.ldb + (int64_t)(k0 >> 6) * g.b_kstep + c * 8),
            (__attribute__((address_space(3))) void*)(Bd + blk * 64), 16, 0, SKF_B_AUX);
    };
    auto dma_A = [&](int k0, int buf) {
#pragma unroll
        for (int p = 0; p < PWA; ++p) dma_A1(k0, buf, p);
    };
    auto dma_B = [&](int k0, int buf) {
#pragma unroll
        for (int p = 0; p < PWB; ++p) dma_B1(k0, buf, p);
    };

    // ABITS: this thread's 32 bits of a tile -- P form: row tid >> 1, k half tid & 1; Q form: k row tid >> 3, 32 columns
    const uint8_t* Abits = (const uint8_t*)g.A;
    auto load_bits = [&](int k0) -> uint32_t {
        if constexpr (AT) {
            int64_t off = (int64_t)(bm0 >> 3) + 4 * (tid & 7);
            if (off > g.lda - 4) off = g.lda - 4;
            return global_load_u32(Abits + (int64_t)(k0 + (tid >> 3)) * g.lda + off);
        } else {
            const int m = bm0 + (tid >> 1);
            const int mc = m < g.M ? m : g.M - 1;
            return global_load_u32(Abits + (int64_t)mc * g.lda + (k0 >> 3) + 4 * (tid & 1));
        }
    };
    auto expand_bits = [&](uint32_t w, int buf, int q) {       // chunk q (8 entries) of the thread's 32
        u32x4* Ad = smem + buf * ASZ;
        const u32x4 v = bits_to_bf16x8((w >> (8 * q)) & 0xFFu);
        if constexpr (AT) {
            const int kr = tid >> 3, c = 4 * (tid & 7) + q;
            lds_write_b128(Ad + kr * 32 + (c ^ (at_key(kr) << 1)), v);
        } else {
            lds_write_b128(Ad + swz_chunk(tid >> 1, 4 * (tid & 1) + q), v);
        }
    };

    {
        // ---- fragment-pipelined K loop --------------------------------------------------------------------------
        // A K tile is consumed in two PHASES (k steps of 32).  While the matrix cores work through the phase's
        // fragments column by column (4 MFMAs per 16-wide column of the wave's sub-tile), the fragments of the NEXT
        // phase are read into the registers the MFMAs have just released: B fragment j+1 behind column j (one spare
        // register set), the four A fragments behind their last MFMA in the last column.  The reads are inline
        // assembly and the loop counts its own lgkmcnt: next phase's first column waits for A fragment i with
        // exactly the later reads outstanding; every other column starts without a wait.  (Round 2, before:
        // all 12 ds_read_b128 of a k step, s_waitcnt lgkmcnt(0), then 32 MFMAs -- with the 8 waves of the workgroup
        // released by the same barrier the LDS needs ~400 cycles for that burst, twice per K tile, against 2048
        // cycles of MFMA work per SIMD.)
        // The workgroup meets ONCE per K tile, between the two phases, with all of its fragment reads complete: every wave
        // then holds the tile's last fragments in registers, so the B buffer of the tile is refilled in phase 1 and its A
        // buffer in phase 0 of the next tile.
        //   phase 0 of tile kt: MFMAs of (kt, k step 0) | reads (kt, k step 1)   | DMA A(kt+2) -> ring slot (kt+2) % 3
        //   s_waitcnt vmcnt: tile kt+1 landed (only A(kt+2) may be in flight); s_barrier
        //   phase 1 of tile kt: MFMAs of (kt, k step 1) | reads (kt+1, k step 0) | DMA B(kt+2) -> ring slot (kt+2) % BST
        constexpr int APH = AT ? 2 : 1;                                  // LDS reads per A fragment
        // lgkmcnt(0) at the barrier: every fragment read a wave has issued is complete when it arrives, so whatever is
        // refilled behind the barrier has no reader left (leaving the last four A fragments in flight measured the same:
        // P12 2.37 / 2.39 vs 2.41 / 2.39 ms)
        constexpr int MIDW = 0x0070;
        const unsigned char* smem_b = (const unsigned char*)smem;
        const int l15 = lane & 15, grp = lane >> 4;
        int aoff[2] = {0, 0}, boff[2], atoff[4] = {0, 0, 0, 0};
        if constexpr (AT) {
            // k rows kr (first half of the fragment) and kr + 4 (+2048 bytes); k step 1 = +32 rows = +16384 bytes
            const int kr0 = 8 * grp + (l15 >> 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = ((wm0 + i * 16) >> 3) + ((l15 & 3) >> 1);
                atoff[i] = kr0 * 512 + ((c ^ (at_key(kr0) << 1)) << 4) + ((l15 & 1) << 3);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) aoff[ks] = swz_chunk(wm0 + l15, 4 * ks + grp) * 16;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) boff[ks] = swz_chunk(wn0 + l15, 4 * ks + grp) * 16;

        u32x4 fa[4], fb[NJ];
        s16x4 fh[4][2];
        auto issue_a = [&](auto ic, auto ksc, const unsigned char* abuf, u32x4& a, s16x4(&h)[2]) {
            constexpr int i = decltype(ic)::value, ks = decltype(ksc)::value;
            if constexpr (AT) {
                h[0] = lds_read_tr16_b64<ks * 16384>(abuf + atoff[i]);
                h[1] = lds_read_tr16_b64<ks * 16384 + 2048>(abuf + atoff[i]);
            } else {
                a = lds_read_b128<i * 2048>(abuf + aoff[ks]);
            }
        };
        auto slot = [&](auto ksc, auto sc, int kt) {
            constexpr int ks = decltype(ksc)::value, sidx = decltype(sc)::value;
            if (kt + 2 >= nkt) return;
            const int k2 = kz0 + (kt + 2) * BK;
            if constexpr (ks == 0) {
                if constexpr (!ABITS && POL::stream_a) dma_A1(k2, (kt + 2) % AST, sidx);
            } else if constexpr (sidx < PWB && POL::stream_b) {
                dma_B1(k2, (kt + 2) % BST, sidx);
            }
        };
        // ABITS, phase 0: chunk q of the bits of tile kt+2 (in wcur since the previous tile) goes into ring slot
        // (kt+2) % 3 behind column min(q, NJ-2) -- every expansion write precedes the phase's A fragment reads, so the
        // lgkmcnt bookkeeping of the fragments is the same with and without them -- then the bits of tile kt+3 are
        // fetched.  Loads and stores are inline assembly like the fragment reads (the compiler's own waitcnt pass put
        // s_waitcnt vmcnt(0) -- every LDS-DMA in flight -- in front of each ds_write of the round-2 loop).
        // Past the end of the K range the writes still go out (stale bits into a ring slot nobody reads again).
        uint32_t wcur = 0u;
        auto expand_chunk = [&](auto qc, int kt) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q == 0) {
                // behind the bits load: the B pieces of tile kt+1 (phase 1 of the previous tile)
                if (kt + 2 < nkt) vm_wait<PWB>(wcur);
            }
            expand_bits(wcur, (kt + 2) % AST, q);
            if constexpr (q == 3) {
                if (kt + 3 < nkt) wcur = load_bits(kz0 + (kt + 3) * BK);
            }
        };
        auto phase = [&](auto ksc, int kt) {
            constexpr int ks = decltype(ksc)::value;
            using NKS = std::integral_constant<int, (ks ^ 1)>;
            const int tn = kt + ks;                                      // the tile this phase reads fragments of
            const unsigned char* abuf = smem_b + (tn % AST) * (ASZ * 16);
            const unsigned char* pb = smem_b + (AST * ASZ + (tn % BST) * BSZ) * 16 + boff[ks ^ 1];
            u32x4 na[4], nb[NJ];
            s16x4 nh[4][2];
            nb[0] = lds_read_b128<0>(pb);
            static_for<NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                static_for<4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (j == 0) {
                        // in flight behind A fragment i: the later A fragments and this phase's first B read
                        if constexpr (AT) {
                            lds_wait<APH * (3 - i) + 1>(fh[i][0], fh[i][1]);
                            const s16x8 v = {fh[i][0][0], fh[i][0][1], fh[i][0][2], fh[i][0][3],
                                             fh[i][1][0], fh[i][1][1], fh[i][1][2], fh[i][1][3]};
                            fa[i] = __builtin_bit_cast(u32x4, v);
                        } else {
                            lds_wait<APH * (3 - i) + 1>(fa[i]);
                        }
                        if constexpr (i == 0) {                          // every B fragment is older than A fragment 0
#pragma unroll
                            for (int q = 0; q < NJ; ++q) lds_claim(fb[q]);
                        }
                    }
                    if constexpr (POL::mfma)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                            __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
                    if constexpr (j == 0) __builtin_amdgcn_sched_barrier(0);    // (keeps MFMA i between waits i and i+1)
                    if constexpr (j == NJ - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_a(ic, NKS{}, abuf, na[i], nh[i]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if constexpr (j + 1 < NJ) nb[j + 1] = lds_read_b128<(j + 1) * 2048>(pb);
                if constexpr (ABITS && ks == 0) {
                    if constexpr (j < 4 && j <= NJ - 2) expand_chunk(jc, kt);
                    if constexpr (NJ == 4 && j == 2) expand_chunk(std::integral_constant<int, 3>{}, kt);
                }
                __builtin_amdgcn_sched_barrier(0);
                // 4 DMA slots per phase, behind the FIRST column of every NJ / 4 (round 6: behind the last one -- columns 1, 3, 5, 7
                // of 8 -- the fourth piece went out beside the four A fragment reads that close the phase, in front of the
                // barrier; one column earlier P12 / P23 / Q23 take 1.5 / 1.3 / 0.9 % less and the iteration 1.1 - 1.8 %;
                // all four pieces up front, or at the head of the columns, measured slower: profiles/r06_contraction_dma_schedule.txt)
                constexpr int CPS = NJ / 4;
                if constexpr (j % CPS == 0) {
                    slot(ksc, std::integral_constant<int, j / CPS>{}, kt);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = na[i];
                fh[i][0] = nh[i][0];
                fh[i][1] = nh[i][1];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = nb[j];
        };

        if (nkt > 0) {
            if constexpr (ABITS) {
                // tiles 0 and 1 expanded before the loop (once per workgroup: plain vmcnt(0)), the bits of tile 2 in wcur
                uint32_t w0 = load_bits(kz0), w1 = 0u;
                dma_B(kz0, 0);
                if (nkt > 1) {
                    w1 = load_bits(kz0 + BK);
                    dma_B(kz0 + BK, 1);
                }
                if (nkt > 2) wcur = load_bits(kz0 + 2 * BK);
                vm_wait<0>(w0);
                vm_wait<0>(w1);
                vm_wait<0>(wcur);
#pragma unroll
                for (int q = 0; q < 4; ++q) expand_bits(w0, 0, q);
#pragma unroll
                for (int q = 0; q < 4; ++q) expand_bits(w1, 1, q);
                __builtin_amdgcn_s_waitcnt(0x0070);
            } else {
                dma_A(kz0, 0);
                dma_B(kz0, 0);
                if (nkt > 1) {
                    dma_A(kz0 + BK, 1);
                    dma_B(kz0 + BK, 1);
                    __builtin_amdgcn_s_waitcnt(0x0F70 | (PWA + PWB));    // tile 0 has landed
                } else {
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {   // fragments of (tile 0, k step 0), in the order every phase issues them: B 0 .. NJ-1, A 0 .. 3
                const unsigned char* pb0 = smem_b + AST * ASZ * 16 + boff[0];
                static_for<NJ>([&](auto jc) { fb[decltype(jc)::value] = lds_read_b128<decltype(jc)::value * 2048>(pb0); });
                static_for<4>([&](auto ic) {
                    issue_a(ic, std::integral_constant<int, 0>{}, smem_b, fa[decltype(ic)::value], fh[decltype(ic)::value]);
                });
            }
            for (int kt = 0; kt < nkt; ++kt) {
                phase(std::integral_constant<int, 0>{}, kt);
                if constexpr (ABITS) {           // B(kt+1) landed; behind it only the bits load of tile kt+3
                    if (kt + 3 < nkt) __builtin_amdgcn_s_waitcnt(MIDW | 1);
                    else __builtin_amdgcn_s_waitcnt(MIDW);
                } else {
                    if (kt + 2 < nkt) __builtin_amdgcn_s_waitcnt(MIDW | PWA);
                    else __builtin_amdgcn_s_waitcnt(MIDW);
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                phase(std::integral_constant<int, 1>{}, kt);
            }
            // the reads of the phase past the end (never consumed) must not land in registers that are live again
            if constexpr (AT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) lds_wait<0>(fh[i][0], fh[i][1]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) lds_wait<0>(fa[i]);
            }
#pragma unroll
            for (int q = 0; q < NJ; ++q) lds_claim(fb[q]);
            if constexpr (ABITS) vm_wait<0>(wcur);
            if constexpr (EPI != EPI_T_STORE) {                          // the epilogues below stage their tile in the rings
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
    }

    // D reg r of a 16 x 16 tile -> row = 4*(lane>>4) + r, col = lane & 15
    if constexpr (EPI == EPI_T_STORE) {
        float* out = (gridDim.z > 1) ? g.part + (int64_t)blockIdx.z * g.M * g.N : g.C;
        const int64_t ldo = (gridDim.z > 1) ? g.N : g.ldc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = bm0 + wm0 + i * 16 + 4 * (lane >> 4) + r;
                    const int n = bn0 + wn0 + j * 16 + (lane & 15);
                    if (m < g.M && n < g.N) out[(int64_t)m * ldo + n] = acc[i][j][r];
                }
    } else {
        bf16_tile_epilogue<EPI, BM, BN, 512, NJ>(g, acc, smem, bm0, bn0, tile, kq0, kq1, kpre);
    }
}

// split-K second stage of the bf16 contraction: C = sum_z part[z]  (fixed order)
static __global__ __launch_bounds__(256) void bf16_splitk_reduce_kernel(float* __restrict__ C, int64_t ldc,
                                                                 const float* __restrict__ part, int M, int N,
                                                                 int splits) {
    const int64_t total = (int64_t)M * N;
    if (ldc == N && (total & 3) == 0) {
        // the contraction outputs (P, Q: ldc == N): 16 bytes per lane, the slices of a step loaded together
        const int64_t nv = total >> 2;
        const f32x4* p4 = (const f32x4*)part;
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nv; e += (int64_t)gridDim.x * blockDim.x) {
            f32x4 v = __builtin_nontemporal_load(p4 + e);
            int z = 1;
            for (; z + 3 < splits; z += 4) {
                const f32x4 a = __builtin_nontemporal_load(p4 + (int64_t)z * nv + e);
                const f32x4 b = __builtin_nontemporal_load(p4 + (int64_t)(z + 1) * nv + e);
                const f32x4 c = __builtin_nontemporal_load(p4 + (int64_t)(z + 2) * nv + e);
                const f32x4 d = __builtin_nontemporal_load(p4 + (int64_t)(z + 3) * nv + e);
                v += a; v += b; v += c; v += d;
            }
            for (; z < splits; ++z) v += __builtin_nontemporal_load(p4 + (int64_t)z * nv + e);
            ((f32x4*)C)[e] = v;
        }
        return;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += part[(int64_t)z * total + e];
        C[(e / N) * ldc + (e % N)] = v;
    }
}

// second stage of a split-K launch: fixed-order sum over the slices, then the epilogue
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, int splits) {
    if (g.gate != nullptr && *g.gate == 0) return;
    const int64_t total = (int64_t)g.M * g.N;
    const T* part = (const T*)g.part;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        T v = (T)0;
        const int64_t src = sym_source(g, e);
        for (int z = 0; z < splits; ++z) v += part[(int64_t)z * total + src];
        epilogue_store<T>(g, (int)(e / g.N), (int)(e % g.N), v);
    }
}

// the same for MANY slices of a small output (wide c x c products: Gram, W = G_i^T P -- up to 250 slices)
template <typename T>
__device__ __forceinline__ void splitk_reduce_z16_body(const GemmArgs& g, int splits) {
    // 16 elements per workgroup pass, 16 threads per element: thread (zl, el) sums the slices zl, zl + 16, ... (four
    // loads in flight), the 16 sums of an element are added in the order zl = 0 .. 15.  (One thread per element walked
    // up to 250 slices of a wide c x c product one load at a time: 75 us for 16 000 outputs.)
    __shared__ T red[16][17];
    if (g.gate != nullptr && *g.gate == 0) return;
    const int64_t total = (int64_t)g.M * g.N;
    const T* part = (const T*)g.part;
    const int zl = threadIdx.x >> 4, el = threadIdx.x & 15;
    for (int64_t e0 = (int64_t)blockIdx.x * 16; e0 < total; e0 += (int64_t)gridDim.x * 16) {
        const int64_t e = e0 + el;
        T v = (T)0;
        if (e < total) {
            const int64_t src = sym_source(g, e);
            int z = zl;
            for (; z + 48 < splits; z += 64) {
                const T a = part[(int64_t)z * total + src], b = part[(int64_t)(z + 16) * total + src];
                const T c = part[(int64_t)(z + 32) * total + src], d = part[(int64_t)(z + 48) * total + src];
                v += a; v += b; v += c; v += d;
            }
            for (; z < splits; z += 16) v += part[(int64_t)z * total + src];
        }
        red[zl][el] = v;
        __syncthreads();
        if (zl == 0 && e < total) {
            T t = red[0][el];
#pragma unroll
            for (int q = 1; q < 16; ++q) t += red[q][el];
            epilogue_store<T>(g, (int)(e / g.N), (int)(e % g.N), t);
        }
        __syncthreads();
    }
}
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_z16_kernel(GemmArgs g, int splits) {
    splitk_reduce_z16_body<T>(g, splits);
}
// ... of the products of a gemm_mfma_group_kernel launch (blockIdx.y = which; the grid-stride loop of the body makes the
// element -> workgroup assignment irrelevant to the sums)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_z16_group_kernel(GemmGroup m) {
    splitk_reduce_z16_body<T>(m.g[blockIdx.y], m.splits[blockIdx.y]);
}

// out[0] = sum_k part[k] in f64, fixed order (single workgroup)
template <typename T>
__global__ __launch_bounds__(256) void sum_partials_kernel(const T* __restrict__ part, int n, double* out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) s += (double)part[k];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------
// G <- G * sqrt(E / max(D, eps))     reference _dfmf.py:294-296, eps = finfo(float64).eps
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mult_update_kernel(T* __restrict__ G, const T* __restrict__ E,
                                                          const T* __restrict__ D, int64_t rows, int cols,
                                                          int64_t ldg, int64_t lde) {
    const T eps = (T)2.220446049250313e-16;
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cols;
        const int c = (int)(e % cols);
        const T d = D[r * lde + c];
        const T den = (d > eps || d != d) ? d : eps;     // np.maximum(D, eps) (NaN propagates)
        const T q = E[r * lde + c] / den;
        G[r * ldg + c] = G[r * ldg + c] * (T)sqrt(q);
    }
}

// ------------------------------------------------------------------------------------------
// counter-based synthetic data, bit-identical to oracle/dfmf_oracle.py::hash_uniform
// ------------------------------------------------------------------------------------------
__device__ __host__ __forceinline__ double hash_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + idx;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 40) * (1.0 / 16777216.0);
}

template <typename T> __device__ __forceinline__ T from_double(double v) { return (T)v; }
template <> __device__ __forceinline__ uint16_t from_double<uint16_t>(double v) { return f32_to_bf16_rne((float)v); }

// element (r, c) of a rows x cols matrix gets hash_uniform(seed, r*cols + c) * scale + shift
template <typename T>
__global__ __launch_bounds__(256) void fill_uniform_kernel(T* __restrict__ dst, int64_t rows, int64_t cols,
                                                           int64_t ld, uint64_t seed, double scale,
                                                           double shift) {
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cols, c = e % cols;
        dst[r * ld + c] = from_double<T>(hash_uniform(seed, (uint64_t)e) * scale + shift);
    }
}

// dst(r,c) = (TD) src(r,c)   (f64 <-> f32 conversions of factors / relations)
template <typename TD, typename TS>
__global__ __launch_bounds__(256) void cast_kernel(TD* __restrict__ dst, int64_t ldd,
                                                   const TS* __restrict__ src, int64_t lds, int64_t rows,
                                                   int64_t cols) {
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cols, c = e % cols;
        dst[r * ldd + c] = (TD)src[r * lds + c];
    }
}

template <typename TS> __device__ __forceinline__ uint16_t to_bf16(TS v) { return f32_to_bf16_rne((float)v); }
template <> __device__ __forceinline__ uint16_t to_bf16<uint16_t>(uint16_t v) { return v; }

// dst[r][c] = bf16(src[r][c]); dst's padding columns are zeroed beforehand by the caller
template <typename TS>
__global__ __launch_bounds__(256) void to_bf16_kernel(uint16_t* __restrict__ dst, int64_t ldd,
                                                      const TS* __restrict__ src, int64_t lds, int64_t rows,
                                                      int64_t cols) {
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cols, c = e % cols;
        dst[r * ldd + c] = to_bf16<TS>(src[r * lds + c]);
    }
}

// dst[c][r] = bf16(src[r][c])  -- transposed bf16 copy through a 32 x 32 LDS tile (coalesced
// reads and writes).  grid = (ceil(cols/32), ceil(rows/32)), 256 threads.
template <typename TS>
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(uint16_t* __restrict__ dst, int64_t ldd,
                                                                const TS* __restrict__ src, int64_t lds,
                                                                int64_t rows, int64_t cols) {
    __shared__ uint16_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? to_bf16<TS>(src[r * lds + c]) : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < cols && r < rows) dst[c * ldd + r] = tile[tx][ty + 8 * k];
    }
}

// SKF_BF16: the multiplicative update (mult_update_kernel) and the refresh of the stored bf16 G^T in one pass over a
// 32 x 32 tile: G is read and written once, the tile leaves transposed through LDS.  grid = (ceil(c/32), ceil(n/32)).
// `Grow` (optional): the bf16 ROWS of the new factor as well, [rows][ldrow] -- the gathered matrix of the list passes.
static __global__ __launch_bounds__(256) void mult_update_transpose_kernel(float* __restrict__ G, const float* __restrict__ E,
                                                                    const float* __restrict__ D, int64_t rows, int64_t cols,
                                                                    uint16_t* __restrict__ GT, int64_t ldgt,
                                                                    uint16_t* __restrict__ Grow, int64_t ldrow) {
    __shared__ uint16_t tile[32][33];
    const float eps = 2.220446049250313e-16f;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
        uint16_t h = 0;
        if (r < rows && c < cols) {
            const float d = D[r * cols + c];
            const float den = (d > eps || d != d) ? d : eps;     // np.maximum(D, eps) (NaN propagates)
            const float g = G[r * cols + c] * sqrtf(E[r * cols + c] / den);
            G[r * cols + c] = g;
            h = f32_to_bf16_rne(g);
            if (Grow) Grow[r * ldrow + c] = h;
        }
        tile[ty + 8 * k][tx] = h;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < cols && r < rows) GT[c * ldgt + r] = tile[tx][ty + 8 * k];
    }
}

// DFMC iteration 0: R[mask] = 0   (_dfmc.py:287-292); the mask is packed, one bit per entry
template <typename T>
__global__ __launch_bounds__(256) void mask_zero_kernel(T* __restrict__ R, int64_t ldr,
                                                        const uint8_t* __restrict__ mbits, int64_t ldmb,
                                                        int64_t rows, int64_t cols) {
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cols, c = e % cols;
        if ((mbits[r * ldmb + (c >> 3)] >> (c & 7)) & 1) R[r * ldr + c] = (T)0;
    }
}

// mask bytes (one per entry, != 0 = unknown) -> packed bits [rows][ldmb bytes]; bits / bytes past `cols` are zero
static __global__ __launch_bounds__(256) void pack_mask_kernel(uint8_t* __restrict__ dst, int64_t ldmb,
                                                        const uint8_t* __restrict__ src, int64_t lds,
                                                        int64_t rows, int64_t cols) {
    const int64_t total = rows * ldmb;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / ldmb, b = e % ldmb;
        uint32_t v = 0u;
        const uint8_t* p = src + r * lds + b * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (b * 8 + q < cols && p[q]) v |= 1u << q;
        dst[e] = (uint8_t)v;
    }
}

// packed mask rows copied into the engine's layout (bytes past the source row are zeroed by the caller)
static __global__ __launch_bounds__(256) void copy_mask_bits_kernel(uint8_t* __restrict__ dst, int64_t ldmb,
                                                             const uint8_t* __restrict__ src, int64_t lds,
                                                             int64_t rows, int64_t cols) {
    const int64_t nb = (cols + 7) / 8, total = rows * nb;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / nb, b = e % nb;
        uint32_t v = src[r * lds + b];
        const int64_t left = cols - b * 8;
        if (left < 8) v &= (1u << left) - 1u;
        dst[r * ldmb + b] = (uint8_t)v;
    }
}

// A binary relation (every entry 0 or 1, e.g. "movie has genre") as a bitmap: bit (c & 7) of dst[r * ldb + (c >> 3)] =
// (src[r][c] == 1); rows and bytes of the padding are zero.  *bad is set when an entry is neither 0 nor 1.
static __global__ __launch_bounds__(256) void pack_binary_kernel(uint8_t* __restrict__ dst, int64_t ldb, int64_t rows_pad,
                                                          const uint16_t* __restrict__ src, int64_t lds,
                                                          int64_t rows, int64_t cols, int* __restrict__ bad) {
    const int64_t total = rows_pad * ldb;
    int wrong = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / ldb, b = e % ldb;
        uint32_t v = 0u;
        if (r < rows) {
            const uint16_t* p = src + r * lds + b * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (b * 8 + q < cols) {
                    const uint16_t h = p[q];
                    if (h == 0x3F80u) v |= 1u << q;
                    else if (h != 0u && h != 0x8000u) wrong = 1;
                }
        }
        dst[e] = (uint8_t)v;
    }
    if (wrong) *bad = 1;
}

// Known entries of a masked bf16 relation per 256 x 256 tile (the tile grid of gemm_bf16_v2_kernel<.., EPI_T_COMPLETE>:
// blockIdx.x = tile of relation ROWS, blockIdx.y = tile of relation COLUMNS, t = blockIdx.y * gridDim.x + blockIdx.x).
// Pass 0 (list == nullptr) counts them into counts[t]; after the host's prefix sum pass 1 writes the entries
// (m_loc * 256 + n_loc) | bf16 value << 16 of tile t to list[off[t] ...] (order inside a tile is irrelevant).
struct KnownArgs {
    const uint8_t* mbits;    // packed mask [rows][ldmb], 1 = unknown; bits past `cols` are zero
    int64_t ldmb;
    const uint16_t* Rin;     // the caller's bf16 relation [rows][ldin]
    int64_t ldin;
    int rows, cols;
    uint32_t* counts;
    const uint32_t* off;
    uint32_t* list;
    int tile_cols;           // 256 or 128: columns per tile (rows per tile: 256)
};
static __global__ __launch_bounds__(256) void known_entries_kernel(KnownArgs a) {
    __shared__ uint32_t cnt;
    const int tid = threadIdx.x;
    const int wpr = a.tile_cols >> 5;                         // 32-column words per tile row
    const int row0 = blockIdx.x * 256, col0 = blockIdx.y * a.tile_cols;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0) cnt = 0u;
    __syncthreads();
    const uint32_t base = a.list ? a.off[tile] : 0u;
    uint32_t mine = 0u;
    for (int it = tid; it < 256 * wpr; it += 256) {        // (row, 32-column word) items
        const int r = it / wpr, w = it % wpr;
        const int m = row0 + r, c0 = col0 + 32 * w;
        if (m >= a.rows || c0 >= a.cols || (int64_t)(c0 >> 3) + 4 > a.ldmb) continue;
        const uint32_t word = *(const uint32_t*)(a.mbits + (int64_t)m * a.ldmb + (c0 >> 3));
        const int nvalid = a.cols - c0 < 32 ? a.cols - c0 : 32;
        uint32_t known = ~word & (nvalid == 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u));
        if (!a.list) {
            mine += (uint32_t)__popc(known);
            continue;
        }
        while (known) {
            const int b = __ffs(known) - 1;
            known &= known - 1u;
            const uint32_t pos = atomicAdd(&cnt, 1u);
            const uint32_t val = a.Rin[(int64_t)m * a.ldin + c0 + b];
            a.list[base + pos] = (uint32_t)(r * 256 + 32 * w + b) | (val << 16);
        }
    }
    if (!a.list) {
        atomicAdd(&cnt, mine);
        __syncthreads();
        if (tid == 0) a.counts[tile] = cnt;
    }
}

// ------------------------------------------------------------------------------------------
// Fill strategies of Relation.filled() on the device (reference fusion_graph.py:464-510): an entry is UNKNOWN when
// it is not finite or masked; it is replaced by the mean of the known entries (numpy.nanmean semantics: NaN and
// masked entries are skipped, +-inf take part and make the mean infinite) of the whole matrix, of its row or of
// its column -- a row / column without any known entry takes the overall mean -- or by a constant.
// Statistics in f64: stats[0..rows) row sums, [rows..2 rows) row counts, then column sums / counts, then the
// total sum and count.  Fixed summation order (no float atomics): run-to-run deterministic.
// ------------------------------------------------------------------------------------------
enum { FILL_MEAN = 0, FILL_ROW_MEAN = 1, FILL_COL_MEAN = 2, FILL_CONST = 3 };

template <typename T>
__device__ __forceinline__ bool fill_known(T v, const uint8_t* mask, int64_t ldm, int64_t r, int64_t c) {
    if (mask && mask[r * ldm + c]) return false;
    return v == v;                               // NaN is skipped; +-inf is a value for the means (numpy.nanmean)
}

template <typename T>
__global__ __launch_bounds__(256) void fill_row_stats_kernel(const T* __restrict__ X, int64_t ld, int64_t rows, int64_t cols,
                                                             const uint8_t* __restrict__ mask, int64_t ldm,
                                                             double* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        double s = 0.0, n = 0.0;
        for (int64_t c = lane; c < cols; c += 64) {
            const T v = X[r * ld + c];
            if (fill_known(v, mask, ldm, r, c)) { s += (double)v; n += 1.0; }
        }
        s = wave_sum(s);
        n = wave_sum(n);
        if (lane == 0) { stats[r] = s; stats[rows + r] = n; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fill_col_stats_kernel(const T* __restrict__ X, int64_t ld, int64_t rows, int64_t cols,
                                                             const uint8_t* __restrict__ mask, int64_t ldm,
                                                             double* __restrict__ stats) {
    double* cs = stats + 2 * rows;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0, n = 0.0;
        for (int64_t r = 0; r < rows; ++r) {
            const T v = X[r * ld + c];
            if (fill_known(v, mask, ldm, r, c)) { s += (double)v; n += 1.0; }
        }
        cs[c] = s;
        cs[cols + c] = n;
    }
}

// total sum / count from the row statistics (one workgroup, fixed order)
static __global__ __launch_bounds__(256) void fill_total_kernel(int64_t rows, int64_t cols, double* __restrict__ stats) {
    __shared__ double ps[256], pn[256];
    double s = 0.0, n = 0.0;
    for (int64_t r = threadIdx.x; r < rows; r += 256) { s += stats[r]; n += stats[rows + r]; }
    ps[threadIdx.x] = s;
    pn[threadIdx.x] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tn = 0.0;
        for (int k = 0; k < 256; ++k) { ts += ps[k]; tn += pn[k]; }
        stats[2 * rows + 2 * cols] = ts;
        stats[2 * rows + 2 * cols + 1] = tn;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fill_apply_kernel(T* __restrict__ X, int64_t ld, int64_t rows, int64_t cols,
                                                         const uint8_t* __restrict__ mask, int64_t ldm,
                                                         const double* __restrict__ stats, int strategy, double value) {
    const int64_t total = rows * cols;
    const double ts = stats[2 * rows + 2 * cols], tn = stats[2 * rows + 2 * cols + 1];
    const double overall = ts / tn;              // 0 / 0 = NaN when nothing is known, as numpy.nanmean
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cols, c = e % cols;
        const T v = X[r * ld + c];
        const bool unknown = (mask && mask[r * ldm + c]) || !(v - v == (T)0);      // masked, NaN or +-inf
        if (!unknown) continue;
        double f = value;
        if (strategy == FILL_MEAN) {
            f = overall;
        } else if (strategy == FILL_ROW_MEAN || strategy == FILL_COL_MEAN) {
            // the line's mean (0 / 0 = NaN without a known entry; inf - inf = NaN); a line without a usable mean takes the
            // overall one -- with a mask every non-finite line mean (the reference masks them: masked_invalid,
            // fusion_graph.py:479-480), without one only NaN (:483)
            f = strategy == FILL_ROW_MEAN ? stats[r] / stats[rows + r] : stats[2 * rows + c] / stats[2 * rows + cols + c];
            const bool unusable = mask ? !(f - f == 0.0) : (f != f);
            if (unusable) f = overall;
        }
        X[r * ld + c] = (T)f;
    }
}

// ------------------------------------------------------------------------------------------
// Sparse constraints.  A constraint matrix Theta (n x n) is usually sparse -- lambda I, a handful of must-link /
// cannot-link pairs per object, dicty's 3 % dense ppi -- while the reference multiplies it as a dense matrix
// (_dfmf.py:284-292: D_i += Theta+ G_i, E_i += Theta- G_i).  With a non-zero bound from the caller
// (skf_theta_desc.nnz) the engine keeps it as CSR, built on the device at bind time (count, host prefix sum, fill),
// and the two products become one pass: a wave per row walks the row's non-zeros, gathers the rows of G and adds
// v G[k] to D (v > 0) or -v G[k] to E (v < 0).  Same arithmetic as the dense split; fixed order within a row.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void theta_row_count_kernel(const T* __restrict__ X, int64_t ld, int64_t n,
                                                              int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        int cnt = 0;
        for (int64_t c = lane; c < n; c += 64) cnt += (X[r * ld + c] != (T)0) ? 1 : 0;
        cnt = wave_sum(cnt);
        if (lane == 0) counts[r] = cnt;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void theta_csr_fill_kernel(const T* __restrict__ X, int64_t ld, int64_t n,
                                                             const int64_t* __restrict__ rowptr, int* __restrict__ cols,
                                                             T* __restrict__ vals) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        int64_t base = rowptr[r];
        for (int64_t c0 = 0; c0 < n; c0 += 64) {
            const int64_t c = c0 + lane;
            const T v = c < n ? X[r * ld + c] : (T)0;
            const bool nz = v != (T)0;
            const unsigned long long m = __ballot(nz ? 1 : 0);
            if (nz) {
                const int before = __popcll(m & ((1ull << lane) - 1ull));
                cols[base + before] = (int)c;
                vals[base + before] = v;
            }
            base += __popcll(m);
        }
    }
}

// entries [a, b) of one CSR row against column j of G: d += v G[k][j] (v > 0), e -= v G[k][j] (v < 0), in list order.
// The entries go 64 at a time: one coalesced load of columns / values (lane z holds entry z), then eight gathers in
// flight, independent of each other and of those loads (a chain of two dependent loads per entry before: the longest
// row -- a hub of dicty's ppi -- set the kernel time).  Whole wave; lanes with j >= c carry zeros.
template <typename T>
__device__ __forceinline__ void theta_row_walk(const int* __restrict__ cols, const T* __restrict__ vals, int64_t a, int64_t b,
                                               const T* __restrict__ G, int c, int j, T& e, T& d) {
    const int lane = threadIdx.x & 63;
    for (int64_t q0 = a; q0 < b; q0 += 64) {
        const int nb = (int)(b - q0 < 64 ? b - q0 : 64);
        const int my_c = lane < nb ? cols[q0 + lane] : 0;
        const T my_v = lane < nb ? vals[q0 + lane] : (T)0;
        for (int z0 = 0; z0 < nb; z0 += 8) {
            T g[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = z0 + u < nb ? z0 + u : nb - 1;
                v[u] = z0 + u < nb ? __shfl(my_v, z, 64) : (T)0;
                const int cz = __shfl(my_c, z, 64);
                g[u] = j < c ? G[(int64_t)cz * c + j] : (T)0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (v[u] > (T)0) d += v[u] * g[u];
                else e -= v[u] * g[u];
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void theta_spmm_kernel(const int64_t* __restrict__ rowptr, const int* __restrict__ cols,
                                                         const T* __restrict__ vals, const T* __restrict__ G,
                                                         T* __restrict__ E, T* __restrict__ D, int64_t n, int c) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        const int64_t a = rowptr[r], b = rowptr[r + 1];
        if (a == b) continue;
        for (int j0 = 0; j0 < c; j0 += 64) {
            const int j = j0 + lane;
            T e = (T)0, d = (T)0;
            theta_row_walk<T>(cols, vals, a, b, G, c, j, e, d);
            if (j < c) {
                E[r * c + j] += e;
                D[r * c + j] += d;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// VERY sparse binary relations (SKF_BF16, SKF_REL_BINARY, at most 1 entry in 256 set -- config 5's movie x actor and
// movie x director at 0.1 %): besides the bitmap the plan keeps the positions of the ones as CSR (rows -> columns) and
// CSC (columns -> rows), both in ascending order, and the two contractions become row gathers of the f32 factor:
//     P[m] = sum over the ones of row m of G_j[k]          Q[n] = sum over the ones of column n of G_i[m]
// nnz * c * 4 bytes of L2 / MALL traffic instead of a pass of the matrix cores over the whole bitmap (40k x 40k x 256:
// 0.72 ms on the bitmap kernel).  The sums run over the f32 masters (the bitmap path rounds G to bf16 first), in a
// fixed order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, (unsigned)off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// counts[r] = number of set bits of bitmap row r (words of 8 bytes; ldb % 8 == 0, padding bits are zero)
static __global__ __launch_bounds__(256) void bits_row_count_kernel(const uint8_t* __restrict__ B, int64_t ldb, int64_t rows,
                                                             int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t nw = ldb >> 3;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const unsigned long long* w = (const unsigned long long*)(B + r * ldb);
        int cnt = 0;
        for (int64_t k = lane; k < nw; k += 64) cnt += __popcll(w[k]);
        cnt = wave_sum(cnt);
        if (lane == 0) counts[r] = cnt;
    }
}

// cols[rowptr[r] ...] = the columns of the ones of row r, ascending
static __global__ __launch_bounds__(256) void bits_csr_fill_kernel(const uint8_t* __restrict__ B, int64_t ldb, int64_t rows,
                                                            const int64_t* __restrict__ rowptr, int* __restrict__ cols) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t nw = ldb >> 3;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const unsigned long long* w = (const unsigned long long*)(B + r * ldb);
        int64_t base = rowptr[r];
        for (int64_t k0 = 0; k0 < nw; k0 += 64) {
            unsigned long long x = (k0 + lane < nw) ? w[k0 + lane] : 0ull;
            const int pc = __popcll(x);
            const int incl = wave_incl_scan(pc);
            int64_t at = base + incl - pc;
            while (x) {
                const int b = __ffsll((long long)x) - 1;
                x &= x - 1ull;
                cols[at++] = (int)((k0 + lane) * 64 + b);
            }
            base += __shfl(incl, 63, 64);
        }
    }
}

// transpose of the CSR pattern: per-column counts, then a fill in arbitrary order, then every column's rows sorted
static __global__ __launch_bounds__(256) void csr_col_count_kernel(const int* __restrict__ cols, int64_t nnz, int* __restrict__ colcnt) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&colcnt[cols[q]], 1);
}
static __global__ __launch_bounds__(256) void csr_transpose_fill_kernel(const int64_t* __restrict__ rowptr, const int* __restrict__ cols,
                                                                 int64_t rows, const int64_t* __restrict__ colptr,
                                                                 int* __restrict__ fillpos, int* __restrict__ rowidx) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves)
        for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
            const int c = cols[q];
            rowidx[colptr[c] + atomicAdd(&fillpos[c], 1)] = (int)r;
        }
}
static __global__ __launch_bounds__(256) void csc_sort_kernel(const int64_t* __restrict__ colptr, int* __restrict__ rowidx, int64_t ncols) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * blockDim.x) {
        int* a = rowidx + colptr[c];
        const int n = (int)(colptr[c + 1] - colptr[c]);
        if (n <= 48) {                                   // a column holds a few dozen ones: insertion sort
            for (int i = 1; i < n; ++i) {
                const int v = a[i];
                int j = i - 1;
                for (; j >= 0 && a[j] > v; --j) a[j + 1] = a[j];
                a[j + 1] = v;
            }
            continue;
        }
        // a heavy column (one object related to a large share of the others): heap sort, O(n log n) in place
        auto sift = [&](int root, int end) {
            for (;;) {
                int child = 2 * root + 1;
                if (child >= end) break;
                if (child + 1 < end && a[child] < a[child + 1]) ++child;
                if (a[root] >= a[child]) break;
                const int t = a[root];
                a[root] = a[child];
                a[child] = t;
                root = child;
            }
        };
        for (int i = n / 2 - 1; i >= 0; --i) sift(i, n);
        for (int end = n - 1; end > 0; --end) {
            const int t = a[0];
            a[0] = a[end];
            a[end] = t;
            sift(0, end);
        }
    }
}

// out[r][0 .. c) = sum over q in [ptr[r], ptr[r+1]) of G[idx[q]][0 .. c)      (one wave per output row)
static __global__ __launch_bounds__(256) void binary_spmm_kernel(const int64_t* __restrict__ ptr, const int* __restrict__ idx,
                                                          const float* __restrict__ G, int64_t ldg, float* __restrict__ out,
                                                          int64_t ldo, int64_t rows, int c) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool vec = (c & 3) == 0 && (ldg & 3) == 0 && (ldo & 3) == 0 && ((((uintptr_t)G) | ((uintptr_t)out)) & 15) == 0;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const int64_t a = ptr[r], b = ptr[r + 1];
        if (vec) {
            for (int j0 = 0; j0 < c; j0 += 256) {
                const int j = j0 + 4 * lane;
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
                if (j < c) {
                    int64_t q = a;
                    for (; q + 1 < b; q += 2) {          // two gathers in flight, summed in the order of the list
                        const f32x4 g0 = *(const f32x4*)(G + (int64_t)idx[q] * ldg + j);
                        const f32x4 g1 = *(const f32x4*)(G + (int64_t)idx[q + 1] * ldg + j);
                        s0 += g0;
                        s0 += g1;
                    }
                    if (q < b) s0 += *(const f32x4*)(G + (int64_t)idx[q] * ldg + j);
                    *(f32x4*)(out + r * ldo + j) = s0;
                }
            }
        } else {
            for (int j = lane; j < c; j += 64) {
                float s = 0.f;
                for (int64_t q = a; q < b; ++q) s += G[(int64_t)idx[q] * ldg + j];
                out[r * ldo + j] = s;
            }
        }
    }
}

// flags[0] |= any(Theta > 0), flags[1] |= any(Theta < 0): the all-zero half of a constraint's +- split
// (_dfmf.py:203-208) is never multiplied (e.g. a non-positive similarity has Theta+ == 0)
template <typename T>
__global__ __launch_bounds__(256) void sign_flags_kernel(const T* __restrict__ src, int64_t ld, int64_t rows,
                                                         int64_t cols, int* __restrict__ flags) {
    const int64_t total = rows * cols;
    bool pos = false, neg = false;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const T v = src[(e / cols) * ld + e % cols];
        pos = pos || v > (T)0;
        neg = neg || v < (T)0;
    }
    if (pos) flags[0] = 1;          // benign race: every writer stores the same value
    if (neg) flags[1] = 1;
}

// bf16 engine: dst = bf16(max(src, 0)) (aop = AOP_POS) or bf16(max(-src, 0)) (AOP_NEG); dst is pre-zeroed
static __global__ __launch_bounds__(256) void split_to_bf16_kernel(uint16_t* __restrict__ dst, int64_t ldd,
                                                            const float* __restrict__ src, int64_t lds,
                                                            int64_t rows, int64_t cols, int aop) {
    const int64_t total = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cols, c = e % cols;
        dst[r * ldd + c] = f32_to_bf16_rne(apply_aop(src[r * lds + c], aop));
    }
}

// dst += src  (n x c, both row-major with leading dimension c)
// x[e] *= f   (the null communicator of the rank-emulation runs: a partial sum times the number of ranks stands for the sum)
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* __restrict__ x, int64_t n, T f) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) x[e] *= f;
}

static __global__ __launch_bounds__(256) void add_into_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                       int64_t total) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x)
        dst[e] += src[e];
}

// ------------------------------------------------------------------------------------------
// Symmetric eigen-decomposition by parallel two-sided cyclic Jacobi (f64), one workgroup per
// matrix (blockIdx.x selects the matrix).  A (n x n, row-major, ld = n) is overwritten; V
// receives the eigenvectors as columns; w the eigenvalues.  n must be even (the host pads an
// odd matrix with one decoupled row/column).  Round-robin ("chess tournament") ordering gives
// n/2 disjoint rotation pairs per round and n-1 rounds per sweep.
// Afterwards:  Vs = V * diag(winv),  winv_k = 1/w_k if |w_k| > n_orig * eps * max|w| else 0
// (scipy.linalg.pinv cut-off, rtol = max(M,N)*eps), so that pinv(A) = Vs * V^T.
// ------------------------------------------------------------------------------------------
struct EighArgs {
    double* A;        // [batch] pointers are derived as A + b*stride
    double* V;
    double* Vs;
    double* w;
    int64_t stride;   // elements between consecutive matrices in A / V / Vs
    int64_t wstride;
    const int* n;     // per-matrix (padded, even) order
    const int* n_orig;
    int* chol_ok;     // per matrix: 1 = the Cholesky fast path produced the inverse, 2 = the rank-revealing deflation
                      // did (result in the eigen format Vs, V); 0 = left to the Jacobi eigen-solver
    int max_sweeps;
};

constexpr int EIGH_THREADS = 512;
constexpr int EIGH_MAXN = 1024;

__device__ __forceinline__ void jacobi_pair(int round, int k, int n, int& p, int& q) {
    // players 0..n-1, player n-1 fixed, the others rotate
    const int m = n - 1;
    int a, b;
    if (k == 0) {
        a = n - 1;
        b = round % m;
    } else {
        a = (round + k) % m;
        b = (round - k + m) % m;
    }
    p = a < b ? a : b;
    q = a < b ? b : a;
}

__device__ __forceinline__ void jacobi_eigh_body(const EighArgs& e, const int b) {
    __shared__ double cs[EIGH_MAXN / 2], sn[EIGH_MAXN / 2];
    __shared__ int pp[EIGH_MAXN / 2], qq[EIGH_MAXN / 2];
    __shared__ double red[EIGH_THREADS / 64];
    __shared__ double s_off, s_diag;
    __shared__ int s_rot;                     // some pair rotated in the current sweep
    if (e.chol_ok[b]) return;                 // uniform: fast path already inverted this matrix
    const int n = e.n[b];
    const int tid = threadIdx.x, nt = blockDim.x;
    double* A = e.A + (int64_t)b * e.stride;
    double* V = e.V + (int64_t)b * e.stride;
    double* Vs = e.Vs + (int64_t)b * e.stride;
    double* w = e.w + (int64_t)b * e.wstride;
    const int half = n / 2;

    // symmetrise, V = I
    for (int idx = tid; idx < n * n; idx += nt) {
        const int r = idx / n, c = idx % n;
        V[idx] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nt) {
        const int r = idx / n, c = idx % n;
        if (r < c) {
            const double s = 0.5 * (A[r * n + c] + A[c * n + r]);
            A[r * n + c] = s;
            A[c * n + r] = s;
        }
    }
    __syncthreads();

    for (int sweep = 0; sweep < e.max_sweeps; ++sweep) {
        // convergence: off-diagonal Frobenius mass relative to the diagonal
        double off = 0.0, dg = 0.0;
        for (int idx = tid; idx < n * n; idx += nt) {
            const int r = idx / n, c = idx % n;
            const double v = A[idx];
            if (r == c) dg += v * v; else off += v * v;
        }
        off = wave_sum(off);
        dg = wave_sum(dg);
        if ((tid & 63) == 0) red[tid >> 6] = off;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < nt / 64; ++i) s += red[i];
            s_off = s;
        }
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = dg;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < nt / 64; ++i) s += red[i];
            s_diag = s;
            s_rot = 0;
        }
        __syncthreads();
        if (s_off <= 1e-30 * s_diag || s_off == 0.0) break;       // uniform across the block
        const double s_dnorm = sqrt(s_diag);

        for (int round = 0; round < n - 1; ++round) {
            // phase 1: rotation angles of the n/2 disjoint pairs
            for (int k = tid; k < half; k += nt) {
                int p, q;
                jacobi_pair(round, k, n, p, q);
                const double app = A[p * n + p], aqq = A[q * n + q], apq = A[p * n + q];
                double c = 1.0, s = 0.0;
                // threshold Jacobi: an off-diagonal entry below 1e-2 * eps of the diagonal's norm is left
                // alone.  That is 100x under the absolute accuracy eps * ||A|| of the LAPACK SVD behind
                // scipy.linalg.pinv and far under the cut-off n * eps * sigma_max; it is what lets a
                // rank-deficient Gram matrix converge: its null-space block consists of rounding noise
                // at eps * ||A|| that cyclic rotations never bring to exactly zero (30 sweeps before,
                // now the usual 8-10).  Pairs that do not rotate move no data.
                if (apq != 0.0 && fabs(apq) > 2.2e-18 * s_dnorm) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                    if (s != 0.0) s_rot = 1;      // benign race: every writer stores 1
                }
                cs[k] = c; sn[k] = s; pp[k] = p; qq[k] = q;
            }
            __syncthreads();
            // phase 2: columns  A <- A J,  V <- V J   (row r, pair k)
            for (int idx = tid; idx < n * half; idx += nt) {
                const int r = idx / half, k = idx % half;
                const int p = pp[k], q = qq[k];
                const double c = cs[k], s = sn[k];
                if (s == 0.0) continue;
                const double arp = A[r * n + p], arq = A[r * n + q];
                A[r * n + p] = c * arp - s * arq;
                A[r * n + q] = s * arp + c * arq;
                const double vrp = V[r * n + p], vrq = V[r * n + q];
                V[r * n + p] = c * vrp - s * vrq;
                V[r * n + q] = s * vrp + c * vrq;
            }
            __syncthreads();
            // phase 3: rows  A <- J^T A   (pair k, column col)
            for (int idx = tid; idx < half * n; idx += nt) {
                const int k = idx / n, col = idx % n;
                const int p = pp[k], q = qq[k];
                const double c = cs[k], s = sn[k];
                if (s == 0.0) continue;
                const double apc = A[p * n + col], aqc = A[q * n + col];
                A[p * n + col] = c * apc - s * aqc;
                A[q * n + col] = s * apc + c * aqc;
            }
            __syncthreads();
        }
        if (!s_rot) break;                       // a whole sweep without a rotation: converged (uniform)
    }

    // eigenvalues, cut-off, scaled eigenvectors
    double mx = 0.0;
    for (int k = tid; k < n; k += nt) {
        const double v = A[k * n + k];
        w[k] = v;
        mx = fmax(mx, fabs(v));
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < nt / 64; ++i) s = fmax(s, red[i]);
        s_off = s;
    }
    __syncthreads();
    const double thr = (double)e.n_orig[b] * 2.220446049250313e-16 * s_off;
    for (int idx = tid; idx < n * n; idx += nt) {
        const int c = idx % n;
        const double wc = w[c];
        const double inv = (fabs(wc) > thr) ? 1.0 / wc : 0.0;
        Vs[idx] = V[idx] * inv;
    }
}
static __global__ __launch_bounds__(EIGH_THREADS) void jacobi_eigh_kernel(EighArgs e) { jacobi_eigh_body(e, blockIdx.x); }

// ------------------------------------------------------------------------------------------
// Fast path of the pseudo-inverse: a symmetric positive definite Gram matrix whose pivots stay
// above rel_thr * max(diag) is inverted through its Cholesky factor (pinv == inverse when no
// singular value falls under the cut-off).  One workgroup per matrix:
//   L L^T = A   right-looking, column k staged in LDS, 2 barriers per column
//   X = L^-1    one thread per column (forward substitution)
// and chol_unpack_kernel forms K = X^T X on the whole grid.  A failed pivot test sets
// chol_ok[b] = 0 and leaves the matrix to the Jacobi eigen-solver above (rank-deficient /
// severely ill-conditioned Gram matrices, reference tests/test_n_run.py:14).
// Scratch: L lives in e.Vs, X in e.V (both are only written by the Jacobi path afterwards).
// ------------------------------------------------------------------------------------------
// Pivot test of the fast path.  Cholesky is invariant under diagonal scaling, so a pivot is judged
// against ITS OWN diagonal entry: p_k / a_kk = sin^2 of the angle between factor column k and the span
// of the columns before it -- below rel_thr the columns are (nearly) dependent and the exact
// pseudo-inverse semantics of the eigen path are needed.  A mere difference in scale between the
// columns (a latent dimension 1e-5 times smaller than the largest) stays on the fast path; only
// a diagonal entry so small that scipy.linalg.pinv's cut-off (n * eps * sigma_max, sigma_max <= n *
// max diag) could truncate its direction is handed to the eigen path.
__device__ __forceinline__ double chol_diag_floor(int n) { return (double)n * (double)n * 2.220446049250313e-16; }

static __global__ __launch_bounds__(EIGH_THREADS) void chol_inverse_kernel(EighArgs e, double rel_thr) {
    __shared__ double col[EIGH_MAXN];
    __shared__ double red[EIGH_THREADS / 64];
    __shared__ double s_max;
    const int b = blockIdx.x;
    const int n = e.n_orig[b], ld = e.n[b];
    const int tid = threadIdx.x, nt = blockDim.x;
    const double* A = e.A + (int64_t)b * e.stride;
    double* L = e.Vs + (int64_t)b * e.stride;
    double* X = e.V + (int64_t)b * e.stride;

    double mx = 0.0;
    for (int idx = tid; idx < n * n; idx += nt) {
        const int r = idx / n, c = idx % n;
        const double v = 0.5 * (A[r * ld + c] + A[c * ld + r]);
        L[r * ld + c] = v;
        X[r * ld + c] = 0.0;
        if (r == c) mx = fmax(mx, fabs(v));
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < nt / 64; ++i) s = fmax(s, red[i]);
        s_max = s;
    }
    __syncthreads();
    const double floor_ = chol_diag_floor(n) * s_max;

    for (int k = 0; k < n; ++k) {
        const double piv = L[k * ld + k];
        const double akk = A[k * ld + k];
        if (!(akk > floor_) || !(piv > rel_thr * akk) || !(piv > 0.0)) {   // uniform: every thread reads the same words
            if (tid == 0) e.chol_ok[b] = 0;
            return;
        }
        const double d = sqrt(piv);
        for (int i = k + tid; i < n; i += nt) col[i] = (i == k) ? d : L[i * ld + k] / d;
        __syncthreads();
        // write the finished column and update the trailing lower triangle
        const int m = n - k - 1;
        for (int i = k + tid; i < n; i += nt) L[i * ld + k] = col[i];
        for (int idx = tid; idx < m * m; idx += nt) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) L[i * ld + j] -= col[i] * col[j];
        }
        __syncthreads();
    }
    // X = L^-1 by forward substitution of L X = I: thread j owns column j and never reads another
    // thread's data, so no barrier is needed.  The loops run over uniform bounds (q < i for every
    // lane; the structurally zero X(q,j), q < j, are simply multiplied in): L(i,q) is then a
    // wave-uniform (scalar) load and the X column loads are coalesced and pipeline freely.
    __syncthreads();
    for (int j = tid; j < n; j += nt) {
        for (int i = 0; i < n; ++i) {
            const double* Li = L + i * ld;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int q = 0;
            for (; q + 3 < i; q += 4) {
                s0 += Li[q] * X[q * ld + j];
                s1 += Li[q + 1] * X[(q + 1) * ld + j];
                s2 += Li[q + 2] * X[(q + 2) * ld + j];
                s3 += Li[q + 3] * X[(q + 3) * ld + j];
            }
            for (; q < i; ++q) s0 += Li[q] * X[q * ld + j];
            const double rhs = (i == j) ? 1.0 : 0.0;
            X[i * ld + j] = (rhs - ((s0 + s1) + (s2 + s3))) / Li[i];
        }
    }
    if (tid == 0) e.chol_ok[b] = 1;
}

// ------------------------------------------------------------------------------------------
// Small orders (n <= 64: the ranks of the reference's own examples): one 64-lane workgroup per
// matrix, everything in LDS, lane i owns row i of L and column i of X = L^-1 (stored transposed,
// so both are bank-conflict-free private rows of pitch 65).  Same contract as chol_inverse_kernel
// (X in e.V, verdict in chol_ok); ~10 us instead of ~100 us for the blocked kernel on a 50 x 50
// matrix -- on small graphs the pseudo-inverse chain is the critical path of an iteration.
// ------------------------------------------------------------------------------------------
constexpr int CHOLS_MAXN = 64;

static __global__ __launch_bounds__(64) void chol_inverse_small_kernel(EighArgs e, double rel_thr) {
    constexpr int LD = CHOLS_MAXN + 1;
    // one array: L in the lower triangle and on the diagonal, X^T strictly above it
    // (M[j][r] = X(r, j) for r > j; X(j, j) = 1 / L(j, j) is not stored)
    __shared__ double M[CHOLS_MAXN * LD];
    __shared__ double col[CHOLS_MAXN];
    const int b = blockIdx.x;
    const int n = e.n_orig[b], ld = e.n[b];
    const int i = threadIdx.x;
    const double* A = e.A + (int64_t)b * e.stride;
    double* X = e.V + (int64_t)b * e.stride;

    double mx = 0.0;
    if (i < n) {
        for (int c = 0; c <= i; ++c) M[i * LD + c] = 0.5 * (A[i * ld + c] + A[c * ld + i]);
        mx = fabs(M[i * LD + i]);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    const double floor_ = chol_diag_floor(n) * mx;
    __syncthreads();

    for (int k = 0; k < n; ++k) {
        const double piv = M[k * LD + k];
        const double akk = A[k * ld + k];
        if (!(akk > floor_) || !(piv > rel_thr * akk) || !(piv > 0.0)) {   // uniform
            if (i == 0) e.chol_ok[b] = 0;
            return;
        }
        const double d = sqrt(piv);
        double lik = 0.0;
        if (i >= k && i < n) {
            lik = (i == k) ? d : M[i * LD + k] / d;
            col[i] = lik;
        }
        __syncthreads();
        if (i >= k && i < n) {
            M[i * LD + k] = lik;
            for (int j = k + 1; j <= i; ++j) M[i * LD + j] -= lik * col[j];
        }
        __syncthreads();
    }
    // X = L^-1: lane j solves L x = e_j (column j of X, kept as row j above the diagonal); the L rows
    // are wave-uniform broadcasts, the x entries lane-private
    if (i < n) {
        const int j = i;
        const double xjj = 1.0 / M[j * LD + j];
        for (int r = j + 1; r < n; ++r) {
            double s = M[r * LD + j] * xjj;
            for (int q = j + 1; q < r; ++q) s += M[r * LD + q] * M[j * LD + q];
            M[j * LD + r] = -s / M[r * LD + r];
        }
    }
    __syncthreads();
    if (i < n)
        for (int c = 0; c < n; ++c)
            X[i * ld + c] = (c < i) ? M[c * LD + i] : (c == i ? 1.0 / M[i * LD + i] : 0.0);
    if (i == 0) e.chol_ok[b] = 1;
}

// ------------------------------------------------------------------------------------------
// LDS-blocked version of the fast path (orders up to CHOLB_MAXN): the same contract as
// chol_inverse_kernel -- L in e.Vs, X = L^-1 in e.V, verdict in chol_ok -- with NB = 32 column
// panels.  Per panel: the diagonal block is factored in LDS, the rows below are solved one per
// thread against it and kept as a transposed panel in LDS, and the trailing matrix receives one
// rank-32 update (global memory is touched once per panel instead of once per column).
// X = L^-1 by block forward substitution: the diagonal blocks are inverted first, then wave w
// walks down block column w with 32x32x32 register-tiled products.
// Dynamic LDS: D[32][33] + max(P[32][n], 8 waves x T[32][33]) doubles.
// ------------------------------------------------------------------------------------------
constexpr int CHOLB_NB = 32;
constexpr int CHOLB_MAXN = 512;

static __global__ __launch_bounds__(EIGH_THREADS) void chol_inverse_blocked_kernel(EighArgs e, double rel_thr) {
    constexpr int NB = CHOLB_NB;
    HIP_DYNAMIC_SHARED(double, csm)
    __shared__ double red[EIGH_THREADS / 64];
    __shared__ double s_max;
    const int b = blockIdx.x;
    const int n = e.n_orig[b], ld = e.n[b];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
    const double* A = e.A + (int64_t)b * e.stride;
    double* L = e.Vs + (int64_t)b * e.stride;
    double* X = e.V + (int64_t)b * e.stride;
    double* D = csm;                       // [NB][NB+1]
    double* P = csm + NB * (NB + 1);       // [NB][n]  transposed panel  /  per-wave T tiles later

    double mx = 0.0;
    for (int idx = tid; idx < n * n; idx += nt) {
        const int r = idx / n, c = idx % n;
        const double v = 0.5 * (A[r * ld + c] + A[c * ld + r]);
        L[r * ld + c] = v;
        X[r * ld + c] = 0.0;
        if (r == c) mx = fmax(mx, fabs(v));
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < nwaves; ++i) s = fmax(s, red[i]);
        s_max = s;
    }
    __syncthreads();
    const double floor_ = chol_diag_floor(n) * s_max;

    // ---------------- factorisation
    for (int kb = 0; kb < n; kb += NB) {
        const int nb = (n - kb < NB) ? n - kb : NB;
        const int m = n - kb - nb;                       // rows below the diagonal block
        for (int idx = tid; idx < nb * nb; idx += nt) {
            const int r = idx / nb, c = idx % nb;
            D[r * (NB + 1) + c] = L[(kb + r) * ld + kb + c];
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            const double piv = D[k * (NB + 1) + k];
            const double akk = A[(kb + k) * ld + kb + k];
            if (!(akk > floor_) || !(piv > rel_thr * akk) || !(piv > 0.0)) {   // uniform: same words for every thread
                if (tid == 0) e.chol_ok[b] = 0;
                return;
            }
            const double d = sqrt(piv);
            __syncthreads();                              // everybody has read the pivot
            for (int r = k + tid; r < nb; r += nt) D[r * (NB + 1) + k] = (r == k) ? d : D[r * (NB + 1) + k] / d;
            __syncthreads();
            const int w = nb - k - 1;
            for (int idx = tid; idx < w * w; idx += nt) {
                const int r = k + 1 + idx / w, c = k + 1 + idx % w;
                if (c <= r) D[r * (NB + 1) + c] -= D[r * (NB + 1) + k] * D[c * (NB + 1) + k];
            }
            __syncthreads();
        }
        // diagonal block back to global; panel rows: x = a * L11^-T, one row per thread
        for (int idx = tid; idx < nb * nb; idx += nt) {
            const int r = idx / nb, c = idx % nb;
            if (c <= r) L[(kb + r) * ld + kb + c] = D[r * (NB + 1) + c];
        }
        for (int i = tid; i < m; i += nt) {
            double* row = L + (int64_t)(kb + nb + i) * ld + kb;
            double x[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) x[c] = (c < nb) ? row[c] : 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c < nb) {
                    double s = x[c];
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (q < c) s -= x[q] * D[c * (NB + 1) + q];
                    x[c] = s / D[c * (NB + 1) + c];
                }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (c < nb) {
                    row[c] = x[c];
                    P[c * n + i] = x[c];
                }
        }
        __syncthreads();
        // trailing update of the lower triangle: L22 -= L21 L21^T
        for (int idx = tid; idx < m * m; idx += nt) {
            const int i = idx / m, j = idx % m;
            if (j <= i) {
                double s = 0.0;
#pragma unroll 8
                for (int c = 0; c < nb; ++c) s += P[c * n + i] * P[c * n + j];
                L[(int64_t)(kb + nb + i) * ld + kb + nb + j] -= s;
            }
        }
        __syncthreads();
    }

    // ---------------- X = L^-1
    const int nblk = (n + NB - 1) / NB;
    // (a) inverses of the diagonal blocks: 32 threads per block, one column each
    for (int t = tid; t < nblk * NB; t += nt) {
        const int blk = t / NB, j = t % NB;
        const int b0 = blk * NB;
        const int nb = (n - b0 < NB) ? n - b0 : NB;
        if (j < nb) {
            double x[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                x[i] = 0.0;
                if (i < nb && i >= j) {
                    double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (q < i && q >= j) s -= L[(int64_t)(b0 + i) * ld + b0 + q] * x[q];
                    x[i] = s / L[(int64_t)(b0 + i) * ld + b0 + i];
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (i < nb && i >= j) X[(int64_t)(b0 + i) * ld + b0 + j] = x[i];
        }
    }
    __syncthreads();
    // (b) block forward substitution: X[ib,jb] = -Xd[ib] * sum_{kb=jb}^{ib-1} L[ib,kb] X[kb,jb]
    double* T = P + wave * NB * (NB + 1);                 // per-wave 32 x 33 tile
    const int r0 = 4 * (lane >> 3), c0 = 4 * (lane & 7);  // 4 x 4 outputs per lane
    for (int ib = 1; ib < nblk; ++ib) {
        const int i0 = ib * NB;
        for (int jb = wave; jb < ib; jb += nwaves) {
            const int j0 = jb * NB;
            double acc[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
            for (int k = j0; k < i0; ++k) {               // k runs over the columns of L[ib, jb..ib-1]
                double a[4], bb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = (i0 + r0 + u < n) ? L[(int64_t)(i0 + r0 + u) * ld + k] : 0.0;
#pragma unroll
                for (int v = 0; v < 4; ++v) bb[v] = X[(int64_t)k * ld + j0 + c0 + v];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * bb[v];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) T[(r0 + u) * (NB + 1) + c0 + v] = acc[u][v];
            // the tile is produced and consumed by the same wave: order the LDS traffic
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u;
                double o[4] = {0.0, 0.0, 0.0, 0.0};
                if (i0 + r < n) {
                    for (int q = 0; q <= r; ++q) {
                        const double dv = X[(int64_t)(i0 + r) * ld + i0 + q];
#pragma unroll
                        for (int v = 0; v < 4; ++v) o[v] += dv * T[q * (NB + 1) + c0 + v];
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) X[(int64_t)(i0 + r) * ld + j0 + c0 + v] = -o[v];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
    if (tid == 0) e.chol_ok[b] = 1;
}

// ------------------------------------------------------------------------------------------
// Rank-revealing deflation: the pseudo-inverse of a symmetric positive SEMI-definite matrix the Cholesky fast
// path rejected (a rank-deficient Gram matrix: duplicate or zero latent columns, rank > objects -- the case of
// reference tests/test_n_run.py:14), without an eigen-decomposition:
//   1. Cholesky with complete diagonal pivoting, stopped when the largest remaining diagonal entry falls below
//      lo * d_max:   A = L L^T + (remainder <= n * lo * d_max),  L: n x r, r = numerical rank
//   2. A^+ = Y Y^T with Y = L (L^T L)^-1  (exact for a matrix of rank r): B = L^T L (r x r, positive definite),
//      its Cholesky factor, and one forward + backward substitution per row of L.
// scipy.linalg.pinv (reference _dfmf.py:232) cuts singular values <= n * eps * sigma_max instead.  The two agree
// to rounding when the spectrum has a GAP between the kept part and rounding noise, which the kernel checks on
// the pivots it is given: no accepted pivot below hi * d_max (lo = 1e-10, hi = 1e-7 relative to the largest
// diagonal entry).  A pivot inside (lo, hi) * d_max -- a genuinely ill-conditioned matrix -- leaves chol_ok = 0 and
// the Jacobi eigen-solver applies the exact cut-off.  One workgroup per matrix; 1.1 ms instead of 216 ms for a
// rank-128 matrix of order 256 (tools/bench_pinv.py).  Result in the eigen format: Vs = V = Y (row-major, zero
// padded), so that eigh_unpack_pinv forms K = Vs V^T.
// ------------------------------------------------------------------------------------------
constexpr int PCHOL_LDS_R = 176;           // packed r (r + 1) / 2 doubles of the small factor fit the dynamic LDS up to this rank
constexpr int PCHOL_LDS_BYTES = PCHOL_LDS_R * (PCHOL_LDS_R + 1) / 2 * 8;
__device__ __forceinline__ void pchol_pinv_body(const EighArgs& e, const int b, double lo, double hi, int lds_rank) {
    __shared__ double d[EIGH_MAXN];            // remaining diagonal; < 0: the index has been a pivot
    __shared__ double rowk[EIGH_MAXN];         // row of L of the current pivot (its first k entries)
    __shared__ double red_v[EIGH_THREADS / 64];
    __shared__ int red_i[EIGH_THREADS / 64];
    __shared__ double s_val;
    __shared__ int s_idx, s_fail;
    if (e.chol_ok[b] || !(lo < 1.0)) return;   // uniform: the fast path inverted this matrix / deflation switched off
    const int n = e.n[b];
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const double* A = e.A + (int64_t)b * e.stride;
    double* Lt = e.V + (int64_t)b * e.stride;      // Lt[k * n + i] = L[i][k]   (column k contiguous over the rows)
    double* W = e.Vs + (int64_t)b * e.stride;      // B = L^T L, then its Cholesky factor (r x r, row-major, ld n)

    for (int i = tid; i < n; i += nt) d[i] = A[(int64_t)i * n + i];
    if (tid == 0) s_fail = 0;
    __syncthreads();
    double dmax0 = 0.0, last = 0.0;
    int r = 0;
    for (int k = 0; k < n; ++k) {
        // ---- pivot = the largest remaining diagonal entry
        double bv = -1.0;
        int bi = -1;
        for (int i = tid; i < n; i += nt)
            if (d[i] > bv) { bv = d[i]; bi = i; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi >= 0 && (bi < 0 || oi < bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double v = red_v[0];
            int ix = red_i[0];
            for (int w = 1; w < nt / 64; ++w)
                if (red_v[w] > v || (red_v[w] == v && red_i[w] >= 0 && (ix < 0 || red_i[w] < ix))) { v = red_v[w]; ix = red_i[w]; }
            s_val = v;
            s_idx = ix;
        }
        __syncthreads();
        const double pv = s_val;
        const int piv = s_idx;
        if (k == 0) dmax0 = pv;
        if (piv < 0 || !(pv > lo * dmax0) || !(pv > 0.0)) break;          // uniform: everything left is noise
        last = pv;
        r = k + 1;
        const double lkk = sqrt(pv);
        for (int j = tid; j < k; j += nt) rowk[j] = Lt[(int64_t)j * n + piv];
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            double v;
            if (i == piv) {
                v = lkk;
            } else if (d[i] < 0.0) {
                v = 0.0;                                                   // an earlier pivot: above the diagonal
            } else {
                double sacc = A[(int64_t)i * n + piv];
                for (int j = 0; j < k; ++j) sacc -= Lt[(int64_t)j * n + i] * rowk[j];
                v = sacc / lkk;
                const double nd = d[i] - v * v;
                d[i] = nd > 0.0 ? nd : 0.0;
            }
            Lt[(int64_t)k * n + i] = v;
        }
        __syncthreads();
        if (tid == 0) d[piv] = -1.0;
        __syncthreads();
    }
    if (r > 0 && last < hi * dmax0) return;    // uniform: a pivot in the ambiguous band -> exact cut-off (Jacobi)

    // ---- B = L^T L (r x r, lower triangle, packed: element (a, c <= a) at a (a + 1) / 2 + c) -- in LDS when it fits
    // (r <= PCHOL_LDS_R), else in the Vs scratch.  One wave per element: the lanes split the rows (coalesced) and meet
    // in a wave reduction.
    HIP_DYNAMIC_SHARED(double, Cs)
    const bool in_lds = r <= lds_rank;          // the launch reserved lds_rank (lds_rank + 1) / 2 doubles of dynamic LDS
    double* Cp = in_lds ? Cs : W;
    const int nel = r * (r + 1) / 2;
    for (int el = wave; el < nel; el += nt / 64) {
        int a = (int)((sqrt(8.0 * el + 1.0) - 1.0) * 0.5);
        while ((a + 1) * (a + 2) / 2 <= el) ++a;
        while (a * (a + 1) / 2 > el) --a;
        const int c = el - a * (a + 1) / 2;
        double sacc = 0.0;
        for (int i = lane; i < n; i += 64) sacc += Lt[(int64_t)a * n + i] * Lt[(int64_t)c * n + i];
        sacc = wave_sum(sacc);
        if (lane == 0) Cp[el] = sacc;
    }
    __syncthreads();
    // ---- Cholesky of B in place, right-looking
    for (int k = 0; k < r; ++k) {
        const int kk = k * (k + 1) / 2;
        const double pk = Cp[kk + k];
        if (!(pk > 0.0)) {
            if (tid == 0) s_fail = 1;
        }
        __syncthreads();
        if (s_fail) return;                    // uniform (A is untouched: the eigen-solver takes over)
        const double ck = sqrt(pk);
        for (int i = k + 1 + tid; i < r; i += nt) Cp[i * (i + 1) / 2 + k] /= ck;
        __syncthreads();
        if (tid == 0) Cp[kk + k] = ck;
        const int m = r - k - 1;
        for (int idx = tid; idx < m * m; idx += nt) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) Cp[i * (i + 1) / 2 + j] -= Cp[i * (i + 1) / 2 + k] * Cp[j * (j + 1) / 2 + k];
        }
        __syncthreads();
    }
    // ---- Y = L B^-1: per row i of L solve C z = l_i, C^T y = z, in place in column i of Lt (coalesced over i;
    // the factor is read uniformly: LDS broadcast)
    for (int i = tid; i < n; i += nt) {
        for (int k = 0; k < r; ++k) {
            const int kk = k * (k + 1) / 2;
            double s0 = Lt[(int64_t)k * n + i], s1 = 0.0;
            int j = 0;
            for (; j + 1 < k; j += 2) {
                s0 -= Cp[kk + j] * Lt[(int64_t)j * n + i];
                s1 -= Cp[kk + j + 1] * Lt[(int64_t)(j + 1) * n + i];
            }
            if (j < k) s0 -= Cp[kk + j] * Lt[(int64_t)j * n + i];
            Lt[(int64_t)k * n + i] = (s0 + s1) / Cp[kk + k];
        }
        for (int k = r - 1; k >= 0; --k) {
            double s0 = Lt[(int64_t)k * n + i], s1 = 0.0;
            int j = k + 1;
            for (; j + 1 < r; j += 2) {
                s0 -= Cp[j * (j + 1) / 2 + k] * Lt[(int64_t)j * n + i];
                s1 -= Cp[(j + 1) * (j + 2) / 2 + k] * Lt[(int64_t)(j + 1) * n + i];
            }
            if (j < r) s0 -= Cp[j * (j + 1) / 2 + k] * Lt[(int64_t)j * n + i];
            Lt[(int64_t)k * n + i] = (s0 + s1) / Cp[k * (k + 1) / 2 + k];
        }
    }
    __syncthreads();
    // ---- eigen format: Vs = V = Y row-major, zero padded   (W is free now; Lt is read before it is overwritten)
    for (int idx = tid; idx < n * n; idx += nt) {
        const int i = idx / n, k = idx % n;
        W[idx] = k < r ? Lt[(int64_t)k * n + i] : 0.0;
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nt) Lt[idx] = W[idx];
    if (tid == 0) e.chol_ok[b] = 2;
}
static __global__ __launch_bounds__(EIGH_THREADS) void pchol_pinv_kernel(EighArgs e, double lo, double hi, int lds_rank) {
    pchol_pinv_body(e, blockIdx.x, lo, hi, lds_rank);
}

// ------------------------------------------------------------------------------------------
// The same deflation over SEVERAL workgroups, for orders above 256 (round 6; pchol_pinv_kernel takes 19.5 ms at order 512 /
// rank 256 and 146 ms at 1023 / 512 in its one workgroup -- the case reference tests/test_n_run.py:14 constructs: rank >
// objects).  One launch per BLOCK of up to 32 pivots, grid = (matrices, slabs of 64 rows), no grid barrier: a launch reads
// one copy of the remaining diagonal / the state and writes the other, every workgroup repeats the small serial part
// (the 32 largest remaining diagonal entries by rank counting, their 32 x 32 block of the Schur complement, its Cholesky
// factorisation with REJECTION of pivots that fell to the noise level inside the block) and owns the rows of its slab:
//     C[i][p] = A[i][piv_p] - sum_{k < r} L[i][k] L[piv_p][k]      (the lazily evaluated columns, as pchol_pinv_kernel)
//     L[i][r + a] = (C[i][p_a] - sum_{b < a} L[i][r + b] Lb[a][b]) / Lb[a][a]      for the accepted pivots p_0 < p_1 < ...
//     d[i] -= sum_a L[i][r + a]^2
// Same acceptance rule and the same gap test as the one-workgroup kernel (pivot > lo * d_max; an accepted pivot below
// hi * d_max leaves the matrix to the eigen-solver); the CANDIDATES of a block are the 32 largest entries of the diagonal as
// it stood before the block, inside the block the pivots follow the current Schur complement (complete pivoting over the
// candidates): the factor differs from the one-workgroup kernel's, A = L L^T and the pseudo-inverse do not.  The host issues 2 ceil(n / 32) + 1 launches blind -- a launch whose matrix is finished, or was
// inverted by the fast path, returns at once --, then
//     pchol_verdict_kernel   gate[b] = the deflation finished cleanly; n_defl[b] = its order (0: the finishing launches idle)
//     B = L^T L (+ 1 on the diagonal beyond the rank)      gated product, pchol_patch_kernel
//     B^-1                                                  the blocked sweep of the fast path (sweep_step_kernel<BIG>) on B
//     Y^T = B^-1 L^T ,  K = Y Y^T                           gated products, straight into the K slot
//     pchol_done_kernel      chol_ok[b] = 1
// Order 512 / rank 256: see profiles/ (tools/bench_pinv.py).  A matrix the route declines at any point keeps chol_ok = 0 and
// falls to pchol_pinv_kernel / the eigen-solver exactly as before.
// ------------------------------------------------------------------------------------------
constexpr int DEFL_NB = 32, DEFL_ROWS = 64, DEFL_KC = 32, DEFL_THREADS = 256;
struct DeflArgs {
    double* d;         // [batch][2][EIGH_MAXN]  remaining diagonal (< 0: the index has been a pivot), two copies
    double* vals;      // [batch][2][2]          d_max of the input, smallest accepted pivot
    int* state;        // [batch][2][2]          rank so far, done (0 running, 1 finished, 2 declined: ambiguous spectrum)
    int* n_defl;       // [batch]                order for the finishing launches (0 = none)
    int* gate;         // [batch]                1 = finishing products run
    int* ok2;          // [batch]                verdict of the sweep over B
    int* rank;         // [batch]                numerical rank (information; tests)
};

static __global__ __launch_bounds__(256) void pchol_init_kernel(EighArgs e, DeflArgs da) {     // L = 0 for the matrices the fast path declined
    const int b = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) { da.gate[b] = 0; da.n_defl[b] = 0; da.ok2[b] = 0; da.rank[b] = -1; }
    if (e.chol_ok[b] != 0) return;
    const int ld = e.n[b];
    double* Lt = e.V + (int64_t)b * e.stride;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < ld * ld; idx += gridDim.x * blockDim.x) Lt[idx] = 0.0;
}

static __global__ __launch_bounds__(DEFL_THREADS) void pchol_step_kernel(EighArgs e, DeflArgs da, double lo, double hi, int step) {
    __shared__ double sd[EIGH_MAXN];
    __shared__ double Lp[DEFL_KC][DEFL_NB + 1];        // L[piv_p][k0 + kk]
    __shared__ double Ls[DEFL_KC][DEFL_ROWS + 1];      // L[row0 + i][k0 + kk]
    __shared__ double Cp[DEFL_NB][DEFL_NB + 1];        // pivot block, then its Cholesky factor (accepted pivots)
    __shared__ double Cs[DEFL_ROWS][DEFL_NB + 1];      // panel of the slab, then its rows of L
    __shared__ double red[DEFL_THREADS / 64];
    __shared__ int piv[DEFL_NB], ord[DEFL_NB], open_[DEFL_NB], seq[DEFL_NB], s_elig, s_pick;
    const int b = blockIdx.x;
    if (e.chol_ok[b] != 0) return;                                          // (uniform: the fast path inverted this matrix)
    const int n = e.n_orig[b], ld = e.n[b];
    const int row0 = blockIdx.y * DEFL_ROWS;
    if (row0 >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int in = step & 1, out = in ^ 1;
    const double* A = e.A + (int64_t)b * e.stride;
    double* Lt = e.V + (int64_t)b * e.stride;                               // Lt[k * ld + i] = L[i][k]
    const double* din = da.d + ((int64_t)b * 2 + in) * EIGH_MAXN;
    double* dout = da.d + ((int64_t)b * 2 + out) * EIGH_MAXN;
    const int* sin = da.state + (b * 2 + in) * 2;
    int* sout = da.state + (b * 2 + out) * 2;
    const double* vin = da.vals + (b * 2 + in) * 2;
    double* vout = da.vals + (b * 2 + out) * 2;
    int r = 0, done = 0;
    double dmax0 = 0.0, last = __builtin_inf();
    if (step == 0) {
        double mx = 0.0;
        for (int i = tid; i < n; i += DEFL_THREADS) {
            const double v = A[(int64_t)i * ld + i];
            sd[i] = v;
            mx = fmax(mx, v);
        }
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        dmax0 = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    } else {
        r = sin[0];
        done = sin[1];
        dmax0 = vin[0];
        last = vin[1];
        if (!done)
            for (int i = tid; i < n; i += DEFL_THREADS) sd[i] = din[i];
    }
    if (done) {                                                             // (uniform) finished in an earlier launch: hand the state on
        if (blockIdx.y == 0 && tid == 0) { sout[0] = r; sout[1] = done; vout[0] = dmax0; vout[1] = last; }
        return;
    }
    if (tid == 0) s_elig = 0;
    if (tid < DEFL_NB) { piv[tid] = 0; ord[tid] = -1; }
    __syncthreads();
    // ---- the (up to) 32 largest eligible entries of the remaining diagonal, by rank counting (ties: the smaller index first)
    const double thr = lo * dmax0;
    for (int i = tid; i < n; i += DEFL_THREADS) {
        const double v = sd[i];
        if (v > thr && v > 0.0) {
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const double w = sd[j];
                rank += (w > v || (w == v && j < i)) ? 1 : 0;
            }
            if (rank < DEFL_NB) piv[rank] = i;
            atomicAdd(&s_elig, 1);
        }
    }
    __syncthreads();
    const int m = s_elig < DEFL_NB ? s_elig : DEFL_NB;
    if (m == 0) {                                                           // (uniform) nothing left above the noise level: finished
        if (blockIdx.y == 0 && tid == 0) {
            sout[0] = r;
            sout[1] = (r > 0 && last < hi * dmax0) ? 2 : 1;
            vout[0] = dmax0;
            vout[1] = last;
        }
        return;
    }
    // ---- panel of the slab and pivot block, lazily: C = A[:, piv] - L[:, :r] L[piv, :r]^T
    const int tx = tid & 31, ty = tid >> 5;                                // column p = tx; rows ty, ty + 8, ...
    double cs[DEFL_ROWS / 8], cp[DEFL_NB / 8];
    const int pc = tx < m ? piv[tx] : piv[0];
#pragma unroll
    for (int q = 0; q < DEFL_ROWS / 8; ++q) {
        const int i = row0 + ty + 8 * q;
        cs[q] = (i < n && tx < m) ? A[(int64_t)i * ld + pc] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < DEFL_NB / 8; ++q) {
        const int pr = ty + 8 * q;
        cp[q] = (pr < m && tx < m) ? A[(int64_t)piv[pr] * ld + pc] : 0.0;
    }
    for (int k0 = 0; k0 < r; k0 += DEFL_KC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DEFL_KC / 8; ++q) {                             // Lp[kk][p]: kk = ty + 8 q, p = tx
            const int kk = ty + 8 * q;
            Lp[kk][tx] = (k0 + kk < r && tx < m) ? Lt[(int64_t)(k0 + kk) * ld + pc] : 0.0;
        }
        {
            const int i = tid & 63;
#pragma unroll
            for (int q = 0; q < DEFL_KC / 4; ++q) {                         // Ls[kk][i]: kk = (tid >> 6) + 4 q
                const int kk = (tid >> 6) + 4 * q;
                Ls[kk][i] = (k0 + kk < r && row0 + i < n) ? Lt[(int64_t)(k0 + kk) * ld + row0 + i] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < DEFL_KC; ++kk) {
            const double lp = Lp[kk][tx];
#pragma unroll
            for (int q = 0; q < DEFL_ROWS / 8; ++q) cs[q] -= Ls[kk][ty + 8 * q] * lp;
#pragma unroll
            for (int q = 0; q < DEFL_NB / 8; ++q) cp[q] -= Lp[kk][ty + 8 * q] * lp;
        }
    }
#pragma unroll
    for (int q = 0; q < DEFL_ROWS / 8; ++q) Cs[ty + 8 * q][tx] = cs[q];
#pragma unroll
    for (int q = 0; q < DEFL_NB / 8; ++q) Cp[ty + 8 * q][tx] = cp[q];
    __syncthreads();
    // ---- Cholesky of the pivot block with COMPLETE pivoting inside the block: the next pivot is the largest current diagonal
    // entry among the block's open candidates (the candidates were picked by the diagonal as it stood before the block; inside
    // it the order follows the Schur complement, as in the one-workgroup kernel), and what the block's earlier pivots took
    // below the noise level is rejected (it stays an ordinary row; its updated diagonal keeps it from being chosen again).
    // Cp keeps full symmetric storage of the open part; column a-th-pivot of Cp holds that column of the factor.
    int n_acc = 0;
    if (tid < DEFL_NB) { open_[tid] = tid < m ? 1 : 0; seq[tid] = 0; }
    __syncthreads();
    for (int t = 0; t < m; ++t) {
        if (tid == 0) {
            int best = -1;
            double bv = thr;
            for (int q = 0; q < m; ++q)
                if (open_[q] && Cp[q][q] > bv && Cp[q][q] > 0.0) { bv = Cp[q][q]; best = q; }
            s_pick = best;
        }
        __syncthreads();
        const int q0 = s_pick;
        if (q0 < 0) break;                                                  // (uniform) the rest of the block is noise
        const double pv = Cp[q0][q0], lkk = sqrt(pv);
        __syncthreads();
        if (tid < m && tid != q0 && open_[tid]) Cp[tid][q0] /= lkk;
        __syncthreads();
        for (int idx = tid; idx < m * m; idx += DEFL_THREADS) {
            const int q = idx / m, c = idx % m;
            if (q != q0 && c != q0 && open_[q] && open_[c]) Cp[q][c] -= Cp[q][q0] * Cp[c][q0];
        }
        __syncthreads();
        if (tid == 0) { Cp[q0][q0] = lkk; open_[q0] = 0; ord[q0] = n_acc; seq[n_acc] = q0; }
        ++n_acc;
        last = fmin(last, pv);
        __syncthreads();
    }
    __syncthreads();
    // ---- the slab's rows of the new columns of L (column r + a belongs to the a-th accepted pivot, candidate seq[a]), the
    // remaining diagonal
    if (tid < DEFL_ROWS && row0 + tid < n) {
        const int i = row0 + tid;
        const double di = sd[i];
        int own = -1;                                                       // this row is an accepted pivot of the block: its candidate index
        for (int p = 0; p < m; ++p)
            if (piv[p] == i && ord[p] >= 0) own = p;
        double sq = 0.0;
        for (int a = 0; a < n_acc; ++a) {
            const int ca = seq[a];
            double x;
            if (di < 0.0) x = 0.0;                                          // a pivot of an earlier block: above the diagonal
            else if (own >= 0) x = a < ord[own] ? Cp[own][ca] : (a == ord[own] ? Cp[own][own] : 0.0);   // its row of the factor
            else {
                double sacc = Cs[tid][ca];
                for (int bb = 0; bb < a; ++bb) sacc -= Cs[tid][seq[bb]] * Cp[ca][seq[bb]];
                x = sacc / Cp[ca][ca];
            }
            Cs[tid][ca] = x;
            sq += x * x;
            Lt[(int64_t)(r + a) * ld + i] = x;
        }
        double nd = di;
        if (own >= 0) nd = -1.0;
        else if (di >= 0.0) { nd = di - sq; nd = nd > 0.0 ? nd : 0.0; }
        dout[i] = nd;
    }
    if (blockIdx.y == 0 && tid == 0) {
        sout[0] = r + n_acc;
        sout[1] = 0;
        vout[0] = dmax0;
        vout[1] = last;
    }
}

// after the last step: did the deflation finish cleanly?  (final = the copy of the state the last launch wrote)
static __global__ __launch_bounds__(64) void pchol_verdict_kernel(EighArgs e, DeflArgs da, int final_copy) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int* s = da.state + (b * 2 + final_copy) * 2;
    const bool ok = e.chol_ok[b] == 0 && s[1] == 1;
    da.gate[b] = ok ? 1 : 0;
    da.n_defl[b] = ok ? e.n_orig[b] : 0;
    da.rank[b] = e.chol_ok[b] == 0 ? s[0] : -1;
}
// B = L^T L has rank r: 1 on the diagonal beyond it makes the matrix the sweep inverts positive definite (the block beyond r
// is the identity and stays decoupled: L's columns there are zero)
static __global__ __launch_bounds__(256) void pchol_patch_kernel(EighArgs e, DeflArgs da, int final_copy) {
    const int b = blockIdx.x;
    if (!da.gate[b]) return;
    const int n = e.n_orig[b], ld = e.n[b], r = da.state[(b * 2 + final_copy) * 2];
    double* B = e.Vs + (int64_t)b * e.stride;
    for (int k = r + threadIdx.x; k < n; k += blockDim.x) B[(int64_t)k * ld + k] = 1.0;
}
static __global__ __launch_bounds__(64) void pchol_done_kernel(EighArgs e, DeflArgs da) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0 && da.gate[b] && da.ok2[b] == 1) e.chol_ok[b] = 1;
}
// the gate of the products behind the sweep over B: both verdicts
static __global__ __launch_bounds__(64) void pchol_gate2_kernel(DeflArgs da) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) da.gate[b] = (da.gate[b] && da.ok2[b] == 1) ? 1 : 0;
}

// K(r,c) = sum_{k >= max(r,c)} X(k,r) X(k,c)   (inverse from the inverted Cholesky factor)
template <typename T>
__global__ __launch_bounds__(256) void chol_unpack_kernel(T* __restrict__ K, int64_t ldk,
                                                          const double* __restrict__ X, int ld, int n,
                                                          const int* __restrict__ chol_ok) {
    if (chol_ok[0] != 1) return;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
        const int r = idx / n, c = idx % n;
        double s = 0.0;
        for (int k = (r > c ? r : c); k < n; ++k) s += X[k * ld + r] * X[k * ld + c];
        K[(int64_t)r * ldk + c] = (T)s;
    }
}

// pad / unpad helpers for the eigen workspace: dst (f64, n_pad x n_pad) <- src (T, n x n);
// the padding row/column is decoupled (zero off-diagonal, zero diagonal -> eigenvalue 0).
template <typename T>
__global__ __launch_bounds__(256) void eigh_pack_kernel(double* __restrict__ dst, int n_pad,
                                                        const T* __restrict__ src, int64_t lds, int n,
                                                        int* __restrict__ n_pad_out = nullptr,
                                                        int* __restrict__ n_out = nullptr) {
    // (the stand-alone operator has no bind step that could upload the two orders: written here, the call needs neither
    //  host-to-device copies nor a stream synchronisation)
    if (n_pad_out && blockIdx.x == 0 && threadIdx.x == 0) {
        n_pad_out[0] = n_pad;
        n_out[0] = n;
    }
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_pad * n_pad; idx += gridDim.x * blockDim.x) {
        const int r = idx / n_pad, c = idx % n_pad;
        dst[idx] = (r < n && c < n) ? (double)src[(int64_t)r * lds + c] : 0.0;
    }
}

// K(r,c) = sum_k Vs(r,k) * V(c,k), r,c < n  (tiny c x c product, f64 accumulate, cast to T)
template <typename T>
__global__ __launch_bounds__(256) void eigh_unpack_pinv_kernel(T* __restrict__ K, int64_t ldk,
                                                               const double* __restrict__ Vs,
                                                               const double* __restrict__ V, int n_pad, int n,
                                                               const int* __restrict__ chol_ok) {
    if (chol_ok[0] == 1) return;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
        const int r = idx / n, c = idx % n;
        double s = 0.0;
        for (int k = 0; k < n_pad; ++k) s += Vs[r * n_pad + k] * V[c * n_pad + k];
        K[(int64_t)r * ldk + c] = (T)s;
    }
}

// ------------------------------------------------------------------------------------------
// Small ranks (every c <= 64: the reference's own examples).  The c x c algebra of a relation is
// a chain of tiny dependent products; as separate launches they dominate the iteration of a small
// graph.  Two kernels, one workgroup per relation, intermediates in LDS, f64 throughout:
//   backbone_small_kernel   T1 = K_i W ; S = nan_to_num(T1 K_j)                (all relations, one launch)
//   bterms_small_kernel     B = (S Gram_j) S^T, D = S^T (Gram_i S) -> += into the +- sums of the row /
//                           column type (one launch per relation: the sums are shared between relations)
// ------------------------------------------------------------------------------------------
constexpr int SMALLC = 64;
constexpr int CHAIN_MAXB = 16;
struct BackboneBatch {
    const double* Ki[CHAIN_MAXB];
    const double* Kj[CHAIN_MAXB];
    const double* W[CHAIN_MAXB];
    double* S[CHAIN_MAXB];
    int ci[CHAIN_MAXB], cj[CHAIN_MAXB];
};

// dynamic LDS: Ki (ci x ci) | W, then T1 (ci x cj) | Kj (cj x cj) | T1 -- operands are staged with
// coalesced loads first: the dot products then run on LDS latency, not on L2 round trips
static __global__ __launch_bounds__(256) void backbone_small_kernel(BackboneBatch bb) {
    HIP_DYNAMIC_SHARED(double, sm)
    const int b = blockIdx.x, ci = bb.ci[b], cj = bb.cj[b];
    double* Ki = sm;
    double* W = Ki + ci * ci;
    double* Kj = W + ci * cj;
    double* T1 = Kj + cj * cj;
    for (int e = threadIdx.x; e < ci * ci; e += blockDim.x) Ki[e] = bb.Ki[b][e];
    for (int e = threadIdx.x; e < ci * cj; e += blockDim.x) W[e] = bb.W[b][e];
    for (int e = threadIdx.x; e < cj * cj; e += blockDim.x) Kj[e] = bb.Kj[b][e];
    __syncthreads();
    for (int e = threadIdx.x; e < ci * cj; e += blockDim.x) {
        const int a = e / cj, c = e % cj;
        double s = 0.0;
        for (int k = 0; k < ci; ++k) s += Ki[a * ci + k] * W[k * cj + c];
        T1[e] = s;
    }
    __syncthreads();
    double* __restrict__ S = bb.S[b];
    for (int e = threadIdx.x; e < ci * cj; e += blockDim.x) {
        const int a = e / cj, c = e % cj;
        double s = 0.0;
        for (int k = 0; k < cj; ++k) s += T1[a * cj + k] * Kj[k * cj + c];
        S[e] = nan_to_num(s);
    }
}

struct BTermsArgs {
    const double* S;        // ci x cj
    const double* Gram_i;   // ci x ci
    const double* Gram_j;   // cj x cj
    double* Bp_i;           // += max(B, 0), ci x ci     (row type)
    double* Bn_i;           // += max(-B, 0)
    double* Bp_j;           // += max(D, 0), cj x cj     (column type)
    double* Bn_j;
    int ci, cj, nan_to_num;
};

// dynamic LDS: S (ci x cj) | Gram_i (ci x ci) | Gram_j (cj x cj) | U (ci x cj)
static __global__ __launch_bounds__(256) void bterms_small_kernel(BTermsArgs a) {
    HIP_DYNAMIC_SHARED(double, sm)
    const int ci = a.ci, cj = a.cj;
    double* S = sm;
    double* Gi = S + ci * cj;
    double* Gj = Gi + ci * ci;
    double* U = Gj + cj * cj;
    for (int e = threadIdx.x; e < ci * cj; e += blockDim.x) S[e] = a.S[e];
    for (int e = threadIdx.x; e < ci * ci; e += blockDim.x) Gi[e] = a.Gram_i[e];
    for (int e = threadIdx.x; e < cj * cj; e += blockDim.x) Gj[e] = a.Gram_j[e];
    __syncthreads();
    // U = S Gram_j ; B = U S^T                                   (tmp2 of _dfmf.py:260)
    for (int e = threadIdx.x; e < ci * cj; e += blockDim.x) {
        const int r = e / cj, c = e % cj;
        double s = 0.0;
        for (int k = 0; k < cj; ++k) s += S[r * cj + k] * Gj[k * cj + c];
        U[e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ci * ci; e += blockDim.x) {
        const int r = e / ci, c = e % ci;
        double s = 0.0;
        for (int k = 0; k < cj; ++k) s += U[r * cj + k] * S[c * cj + k];
        if (a.nan_to_num) s = nan_to_num(s);
        a.Bp_i[e] += s > 0.0 ? s : 0.0;
        a.Bn_i[e] += s > 0.0 ? 0.0 : -s;
    }
    __syncthreads();
    // U = Gram_i S ; D = S^T U                                   (tmp5 of _dfmf.py:272)
    for (int e = threadIdx.x; e < ci * cj; e += blockDim.x) {
        const int r = e / cj, c = e % cj;
        double s = 0.0;
        for (int k = 0; k < ci; ++k) s += Gi[r * ci + k] * S[k * cj + c];
        U[e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cj * cj; e += blockDim.x) {
        const int r = e / cj, c = e % cj;
        double s = 0.0;
        for (int k = 0; k < ci; ++k) s += S[k * cj + r] * U[k * cj + c];
        if (a.nan_to_num) s = nan_to_num(s);
        a.Bp_j[e] += s > 0.0 ? s : 0.0;
        a.Bn_j[e] += s > 0.0 ? 0.0 : -s;
    }
}

// f32 engines: f32 roundings of several small f64 matrices in one launch (blockIdx.y = entry)
constexpr int CAST_MAXB = 32;
struct CastBatch {
    const double* src[CAST_MAXB];
    float* dst[CAST_MAXB];
    int count[CAST_MAXB];
};
static __global__ __launch_bounds__(256) void cast_batched_kernel(CastBatch cb) {
    const int b = blockIdx.y;
    const double* __restrict__ src = cb.src[b];
    float* __restrict__ dst = cb.dst[b];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < cb.count[b]; e += gridDim.x * blockDim.x) dst[e] = (float)src[e];
}

// Batched forms for the plan (blockIdx.y = matrix): the per-type pack / unpack launches of one
// pseudo-inverse pass collapse into one launch each -- on small graphs an iteration is bounded by
// the number of dependent launches, not by their work.
constexpr int PINV_MAXB = 16;
struct PinvBatch {
    const double* gram[PINV_MAXB];   // c x c, ld = c
    double* K[PINV_MAXB];            // c x c, ld = c
    int c[PINV_MAXB], n_pad[PINV_MAXB];
};

static __global__ __launch_bounds__(256) void eigh_pack_batched_kernel(PinvBatch pb, double* __restrict__ A, int64_t stride) {
    const int b = blockIdx.y, n = pb.c[b], n_pad = pb.n_pad[b];
    double* dst = A + (int64_t)b * stride;
    const double* src = pb.gram[b];
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_pad * n_pad; idx += gridDim.x * blockDim.x) {
        const int r = idx / n_pad, c = idx % n_pad;
        dst[idx] = (r < n && c < n) ? src[(int64_t)r * n + c] : 0.0;
    }
}

static __global__ __launch_bounds__(256) void chol_unpack_batched_kernel(PinvBatch pb, const double* __restrict__ Xall,
                                                                  int64_t stride, const int* __restrict__ chol_ok) {
    const int b = blockIdx.y;
    if (chol_ok[b] != 1) return;
    const int n = pb.c[b], ld = pb.n_pad[b];
    const double* X = Xall + (int64_t)b * stride;
    double* K = pb.K[b];
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
        const int r = idx / n, c = idx % n;
        double s = 0.0;
        for (int k = (r > c ? r : c); k < n; ++k) s += X[k * ld + r] * X[k * ld + c];
        K[(int64_t)r * n + c] = s;
    }
}

// ------------------------------------------------------------------------------------------
// Fast path of the pseudo-inverse for orders 65 .. SWEEP_MAXN (round 4): K = A^-1 of a symmetric positive definite matrix by
// the BLOCKED SWEEP OPERATOR, one workgroup of 512 threads per matrix, straight into the K slot -- no factor, no triangular
// inverse, no X^T X product behind it (chol_inverse_blocked_kernel + chol_unpack: 1.15 + 0.09 ms at order 256, and the
// critical path of a rank of the ownership-sharded iteration and of config 5's pipeline; this kernel: see profiles/).
// Sweeping pivot k of a symmetric M: m_ij -= c_i c_j / m_kk (c = column k), row / column k <- c / m_kk, m_kk <- -1 / m_kk;
// after all n pivots M = -A^-1.  Blocked by NB = 32 pivots: with P = M_pp^-1 of the CURRENT pivot block
//     M_rr -= M_rp P M_pr ,  M_rp <- M_rp P ,  M_pr <- P M_pr ,  M_pp <- -P
// (sweeping a block is sweeping its pivots one after the other).  Per block: the panel C = M[:, block] and the pivot block
// go to LDS; the 32 x 32 block is swept pivot by pivot with ONE barrier per pivot (two elements per thread in registers,
// row k published through a double buffer) -- its pivots are the Schur complements a Cholesky factorisation would take the
// roots of, so the verdict is that of the Cholesky kernels: pivot > rel_thr * a_kk, a_kk above the diagonal floor; a failed
// pivot leaves the matrix (untouched in e.A) to the deflation / eigen-solver --; T = M_rp P (n x 32) and the rank-32 update
// of the whole matrix run on the f64 matrix cores (16 x 16 x 4 tiles, operands from LDS at a pitch of 36 words: conflict-free
// fragments; a wave owns 64 x 64 outputs of the update at a time, its accumulators start from M itself).  On this part the
// vector ALU and the matrix cores run f64 FMAs at the same rate (78.6 TFLOP/s either way: 13.7 us per block on one CU at
// order 256); what the matrix cores save is LDS traffic -- the plain-FMA T read two LDS words per FMA and took 14 of the
// 66 us of a block (time stamps of a probe build, -DSKF_PROBE_STAMPS).  M lives in the plan's eigen scratch (e.V), 0.5 MB:
// L2 resident.  Stand-alone at order 256: 0.53 ms (first version 0.63; Cholesky inverse + unpack 0.93) -- per block: panel
// 2 us, pivot-block sweep 14, T 2.6, update 27, write-back 7.
// ------------------------------------------------------------------------------------------
constexpr int SWEEP_MAXN = 256;
constexpr int SWEEP_NB = 32;
constexpr int SWEEP_LD = SWEEP_NB + 4;      // (row r, k) -> 4 r + k mod 32: the fragments of the matrix-core tiles are conflict-free
constexpr int SWEEP_THREADS = 512;
constexpr int SWEEP_LDS_BYTES = ((2 * SWEEP_MAXN + SWEEP_NB) * SWEEP_LD + 6 * SWEEP_NB + SWEEP_MAXN) * 8;

// The sweep of one 32 x 32 pivot block by ONE wave (round 5, sweep_step_kernel): lane = (column c = lane & 31, half =
// lane >> 5) holds rows 16 half .. 16 half + 15 of its column in registers; row k of the current state travels through LDS
// (double-buffered) and the wave orders its own LDS traffic -- no workgroup barrier per pivot.  (sweep_inverse_kernel spreads
// the block over its 512 threads, two elements each, with one __syncthreads() per pivot: 14 us per block -- barriers and LDS
// round trips, not the 2 k FMAs.)  The chain from one pivot to the next is: element of row k + 1 -> LDS -> every lane ->
// reciprocal -> update; so pivot k updates ROW k + 1 FIRST, the lane that holds the next pivot takes its reciprocal at once,
// row and reciprocal go to LDS, and the other 15 rows of the lane are updated while that round trip is in flight.  The loop
// has no data-dependent branch: a failed pivot clears `ok` and the arithmetic runs on (its results are dropped).  The
// padding of a short block (rows / columns >= nb) is zero on entry and stays +0 under every finite pivot, so nothing masks
// it inside the loop.  The arithmetic of an element is the same expression in the same order as in sweep_inverse_kernel
// (1 / pivot is the same IEEE quotient whichever lane takes it), so the bits are the same.
// rowk: SWEEP_ROWK_WORDS words.  Returns false when a pivot failed its bound (uniform).
constexpr int SWEEP_ROWK = 2 * SWEEP_NB;          // a row of the pivot block and, behind it, the reciprocals of its elements
constexpr int SWEEP_ROWK_WORDS = 3 * SWEEP_ROWK;  // two buffers and one nobody reads
__device__ __forceinline__ bool sweep_pivot_block(const double* __restrict__ Cblk, double* __restrict__ Pv,
                                                  double* __restrict__ rowk, const double* __restrict__ need, int nb, int lane) {
    constexpr int NB = SWEEP_NB, LD = SWEEP_LD, RK = SWEEP_ROWK;
    const int c = lane & 31, half = lane >> 5;
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = 16 * half + q;
        v[q] = (r < nb && c < nb) ? Cblk[r * LD + c] : 0.0;
    }
    // every lane stores what it has of row k + 1 and the reciprocal of it -- the half that does not hold the row into a
    // buffer nobody reads: no branch, and the quotient is scheduled among the updates of the other rows
    {
        double* dst = rowk + (half == 0 ? 0 : 2 * RK);
        dst[c] = v[0];
        dst[NB + c] = 1.0 / v[0];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NB; ++k) {                                       // (a fixed trip count: the registers of v are named)
        if (k < nb) {                                                    // (uniform)
            const double* rk = rowk + (k & 1) * RK;
            const double piv = rk[k];                                    // (every lane reads the same words)
            ok = ok & (piv > need[k]);
            const double d = rk[NB + k];
            const double cc = rk[c];
            double e[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) e[q] = rk[16 * half + q] * d;
            auto swept = [&](int q) -> double {
                const double gen = (c == k) ? e[q] : fma(-e[q], cc, v[q]);
                if (q != (k & 15)) return gen;
                const double piv_row = (c == k) ? -d : cc * d;           // row k itself, in the half that holds it
                return (half == (k >> 4)) ? piv_row : gen;
            };
            const int q1 = (k + 1) & 15;                                 // row k + 1 first: it carries the next pivot
            if (k + 1 < NB) {
                v[q1] = swept(q1);
                double* dst = rowk + (half == ((k + 1) >> 4) ? ((k + 1) & 1) * RK : 2 * RK);
                dst[c] = v[q1];
                dst[NB + c] = 1.0 / v[q1];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (!(k + 1 < NB && q == q1)) v[q] = swept(q);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) Pv[(16 * half + q) * LD + c] = v[q];
    return ok;
}

static __global__ __launch_bounds__(SWEEP_THREADS) void sweep_inverse_kernel(EighArgs e, PinvBatch pb, double rel_thr) {
    constexpr int NB = SWEEP_NB, LD = SWEEP_LD;
    HIP_DYNAMIC_SHARED(double, ssm)
    __shared__ double red[SWEEP_THREADS / 64];
    __shared__ double s_max;
    double* Cs = ssm;                              // [SWEEP_MAXN][LD]  panel C = M[:, kb .. kb + nb)
    double* Ts = Cs + SWEEP_MAXN * LD;             // [SWEEP_MAXN][LD]  T = M_rp P
    double* Pv = Ts + SWEEP_MAXN * LD;             // [NB][LD]          the swept pivot block: -P
    double* rowk = Pv + NB * LD;                   // [2][NB]           row k of the pivot block, double-buffered (6 NB words reserved)
    double* need = rowk + 6 * NB;                  // [SWEEP_MAXN]      the bound pivot k has to exceed
    const int b = blockIdx.x;
    const int n = e.n_orig[b], ld = e.n[b];
    if (n > SWEEP_MAXN) return;                    // (the host sends such plans to chol_inverse_blocked_kernel)
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;      // 32 x 16
    const int lane = tid & 63, wave = tid >> 6;
    const double* A = e.A + (int64_t)b * e.stride;
    double* M = e.V + (int64_t)b * e.stride;

    double mx = 0.0;
    for (int idx = tid; idx < n * n; idx += SWEEP_THREADS) {
        const int r = idx / n, c = idx % n;
        const double v = 0.5 * (A[r * ld + c] + A[c * ld + r]);
        M[r * ld + c] = v;
        if (r == c) mx = fmax(mx, fabs(v));
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < SWEEP_THREADS / 64; ++i) s = fmax(s, red[i]);
        s_max = s;
    }
    __syncthreads();
    {   // the three tests on pivot k -- a_kk > floor, pivot > thr a_kk, pivot > 0 -- as ONE bound (+inf where a_kk fails)
        const double floor_ = chol_diag_floor(n) * s_max;
        for (int k = tid; k < n; k += SWEEP_THREADS) {
            const double akk = A[k * ld + k];
            need[k] = (akk > floor_) ? fmax(rel_thr * akk, 0.0) : __builtin_inf();
        }
    }
    __syncthreads();

#ifdef SKF_PROBE_STAMPS
    long long ph[6] = {0, 0, 0, 0, 0, 0}, t_in = wall_clock64();
#define SKF_STAMP(i) { const long long now_ = wall_clock64(); ph[i] += now_ - t_in; t_in = now_; }
#else
#define SKF_STAMP(i)
#endif
    for (int kb = 0; kb < n; kb += NB) {
        const int nb = (n - kb < NB) ? n - kb : NB;
        SKF_STAMP(0)
        // ---- panel and pivot block to LDS
        for (int i = ty; i < n; i += 16) Cs[i * LD + tx] = (tx < nb) ? M[i * ld + kb + tx] : 0.0;
        __syncthreads();
        SKF_STAMP(1)
        // ---- sweep of the pivot block: elements (r, c) = (ty, tx) and (ty + 16, tx) in registers
        double v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = ty + 16 * h;
            v[h] = (r < nb && tx < nb) ? Cs[(kb + r) * LD + tx] : 0.0;
        }
        if (ty == 0) rowk[tx] = v[0];                                    // row 0
        __syncthreads();
        bool ok = true;
        for (int k = 0; k < nb; ++k) {
            const double* rk = rowk + (k & 1) * NB;
            const double piv = rk[k];
            if (!(piv > need[kb + k])) {                                 // (uniform: every thread reads the same words)
                ok = false;
                break;
            }
            const double d = 1.0 / piv;
            const double cc = rk[tx];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = ty + 16 * h;
                const double cr = rk[r];
                const double nv = (r == k) ? ((tx == k) ? -d : cc * d) : ((tx == k) ? cr * d : fma(-cr * d, cc, v[h]));
                v[h] = (r < nb && tx < nb) ? nv : 0.0;
                if (r == k + 1) rowk[((k + 1) & 1) * NB + tx] = v[h];     // row k + 1 of the swept block, for the next pivot
            }
            __syncthreads();
        }
        if (!ok) {                                                       // (uniform)
            if (tid == 0) e.chol_ok[b] = 0;
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) Pv[(ty + 16 * h) * LD + tx] = v[h];
        __syncthreads();
        SKF_STAMP(2)
        // ---- T = M_rp P = -(C Pv), every row (the pivot rows' entries are not used), on the f64 matrix cores: 16 x 16 tiles,
        // row tiles wave, wave + 8, both column tiles on one A fragment.  (The plain-FMA form read two LDS words per FMA: 4 MB
        // per block, 14 of the 66 us a block took at order 256 -- time stamps of a probe build.)
        {
            typedef Mfma<double> MF;
            const int ntile = (n + 15) >> 4;
            for (int it = wave; it < ntile; it += SWEEP_THREADS / 64) {
                const int row = it * 16 + MF::a_row(lane);
                MF::acc_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k0 = 0; k0 < NB; k0 += MF::KT) {
                    const int kk = k0 + MF::ab_k(lane);
                    const double a = row < n ? Cs[row * LD + kk] : 0.0;
                    acc0 = MF::mma(a, Pv[kk * LD + MF::a_row(lane)], acc0);
                    acc1 = MF::mma(a, Pv[kk * LD + 16 + MF::a_row(lane)], acc1);
                }
#pragma unroll
                for (int r = 0; r < MF::NREG; ++r) {
                    const int i = it * 16 + MF::d_row(lane, r);
                    if (i < n) {
                        Ts[i * LD + MF::d_col(lane)] = -acc0[r];
                        Ts[i * LD + 16 + MF::d_col(lane)] = -acc1[r];
                    }
                }
            }
        }
        __syncthreads();
        SKF_STAMP(3)
        // ---- M_rr -= T C^T outside the pivot rows / columns, on the f64 matrix cores: a wave owns 64 x 64 outputs at a time
        // (4 x 4 tiles on four A and four B fragments per K step: 8 LDS reads for 16 instructions; 8 x 8 outputs per thread
        // in plain FMAs read 16 words per 64 FMAs and ran on the LDS, 27 us per block at order 256)
        {
            typedef Mfma<double> MF;
            const int nblk = (n + 63) >> 6;
            for (int blk = wave; blk < nblk * nblk; blk += SWEEP_THREADS / 64) {
                const int bi = (blk / nblk) * 64, bj = (blk % nblk) * 64;
                // the accumulators start from M itself (the loads fly while the first products run) and take -T C^T
                MF::acc_t acc[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < MF::NREG; ++r) {
                        const int i = bi + 16 * a + MF::d_row(lane, r);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int j = bj + 16 * q + MF::d_col(lane);
                            acc[a][q][r] = (i < n && j < n) ? M[i * ld + j] : 0.0;
                        }
                    }
#pragma unroll 2
                for (int k0 = 0; k0 < NB; k0 += MF::KT) {
                    const int kk = k0 + MF::ab_k(lane);
                    double ta[4], cb[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int i = bi + 16 * a + MF::a_row(lane);
                        ta[a] = i < n ? -Ts[i * LD + kk] : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j = bj + 16 * q + MF::a_row(lane);
                        cb[q] = j < n ? Cs[j * LD + kk] : 0.0;
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[a][q] = MF::mma(ta[a], cb[q], acc[a][q]);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < MF::NREG; ++r) {
                        const int i = bi + 16 * a + MF::d_row(lane, r);
                        if (i >= n || (i >= kb && i < kb + nb)) continue;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int j = bj + 16 * q + MF::d_col(lane);
                            if (j < n && !(j >= kb && j < kb + nb)) M[i * ld + j] = acc[a][q][r];
                        }
                    }
            }
        }
        SKF_STAMP(4)
        // ---- pivot columns <- T (lanes along the columns of the block), pivot block <- -P; pivot rows <- T^T with the lanes
        // along the rows of T (the transposed copy as the mirror of the column loop: one 8-byte store per lane ld apart,
        // 9 of the 66 us)
        for (int i = ty; i < n; i += 16) {
            if (tx >= nb) continue;
            if (i >= kb && i < kb + nb) M[i * ld + kb + tx] = Pv[(i - kb) * LD + tx];
            else M[i * ld + kb + tx] = Ts[i * LD + tx];
        }
        for (int idx = tid; idx < nb * n; idx += SWEEP_THREADS) {
            const int c = idx / n, i = idx % n;
            if (!(i >= kb && i < kb + nb)) M[(kb + c) * ld + i] = Ts[i * LD + c];
        }
        __syncthreads();
    }
    SKF_STAMP(5)
#ifdef SKF_PROBE_STAMPS
    if (tid == 0 && n >= 200)
        printf("sweep_inverse n %d: panel load %lld, pivot block sweep %lld, T %lld, update %lld, write-back + barrier %lld (x10 ns, all blocks)\n", n,
               ph[1], ph[2], ph[3], ph[4], ph[5] + ph[0]);
#endif
    // ---- all pivots swept: M = -A^-1
    double* K = pb.K[b];
    for (int idx = tid; idx < n * n; idx += SWEEP_THREADS) K[idx] = -M[(idx / n) * ld + idx % n];
    if (tid == 0) e.chol_ok[b] = 1;
}

// ------------------------------------------------------------------------------------------
// The same sweep with the rank-32 update of a block step spread over SEVERAL workgroups (round 5): one launch per block
// step, grid = (matrices, row slabs).  A slab is `rs` rows of the matrix (a multiple of 32); its workgroup repeats the small
// serial part of the step -- panel to LDS, the pivot-block sweep by one wave -- and then owns everything that carries one of
// its rows i: T_i = M_ip P, the updated M_ij, the pivot columns (i, p) and, transposed, the pivot rows (p, i); the slab that
// holds the pivot rows also writes -P.  A step READS one copy of the matrix and WRITES the other (e.V / e.Vs alternate; the
// first step reads the symmetrised input itself, the last one writes K = -M), so no workgroup waits for another inside a
// launch and there is no grid barrier to hang on: the order between steps is the stream's.  Every element takes the
// arithmetic of sweep_inverse_kernel in the same order (the accumulators of a tile start from M and run over the 32 pivots
// in steps of four) -- the two kernels return the same bits.  chol_ok[b] carries the state between launches: 3 = steps so
// far accepted every pivot, 0 = a pivot failed (the later steps of that matrix return at once; e.A is untouched for the
// deflation / eigen-solver), 1 after the last step.
// One workgroup per matrix spends 53 us per block at order 256, 27 of them in the update; here a step is the serial part
// plus one 32 x 32 tile per wave, and a launch boundary (profiles/: tools/bench_pinv.py).
// ------------------------------------------------------------------------------------------
constexpr int SWEEP_RUNNING = 3;
// BIG (orders above SWEEP_MAXN, up to EIGH_MAXN): the panel of every row does not fit the LDS -- it holds the pivot rows and
// the rows of the slab only (slabs of exactly 32 rows); the column operands C_j of a tile come from the matrix in memory (L2:
// 8 MB at order 1024), asked for together with the tile's accumulators.  Same arithmetic, element by element.
constexpr int SWEEP_BIG_ROWS = 2 * SWEEP_NB;      // pivot rows, slab rows
constexpr int SWEEP_BIG_LDS_BYTES = ((SWEEP_BIG_ROWS + 2 * SWEEP_NB) * SWEEP_LD + 6 * SWEEP_NB + SWEEP_NB) * 8;

template <bool BIG>
__global__ __launch_bounds__(SWEEP_THREADS) void sweep_step_kernel(EighArgs e, PinvBatch pb, double rel_thr, int step, int rs) {
    constexpr int NB = SWEEP_NB, LD = SWEEP_LD;
    constexpr int CROWS = BIG ? SWEEP_BIG_ROWS : SWEEP_MAXN, TROWS = BIG ? SWEEP_NB : SWEEP_MAXN;
    constexpr int PCN = BIG ? SWEEP_BIG_ROWS / 16 : SWEEP_MAXN / 16;
    typedef Mfma<double> MF;
    HIP_DYNAMIC_SHARED(double, ssm)
    __shared__ double red[SWEEP_THREADS / 64];
    __shared__ int s_ok;
    double* Cs = ssm;                              // panel C = M[:, kb .. kb + nb): every row | BIG: pivot rows, then slab rows
    double* Ts = Cs + CROWS * LD;                  // T = M_rp P (rows of this slab)
    double* Pv = Ts + TROWS * LD;                  // [NB][LD]          the swept pivot block: -P
    double* rowk = Pv + NB * LD;                   // [SWEEP_ROWK_WORDS]
    double* need = rowk + 6 * NB;                  // [NB]              the bounds of this step's pivots
    const int b = blockIdx.x;
    const int n = e.n_orig[b], ld = e.n[b];
    const int kb = step * NB;
    if (n > (BIG ? EIGH_MAXN : SWEEP_MAXN) || kb >= n) return;
    if (BIG) rs = NB;
    const int r0 = blockIdx.y * rs, r1 = (r0 + rs < n) ? r0 + rs : n;
    if (r0 >= n) return;
    const bool first = step == 0, last = kb + NB >= n;
    if (!first && *(volatile const int*)(e.chol_ok + b) == 0) return;     // (uniform; slab 0 of THIS launch may already have written its verdict)
    const int nb = (n - kb < NB) ? n - kb : NB;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;      // 32 x 16
    const int lane = tid & 63, wave = tid >> 6;
    const double* A = e.A + (int64_t)b * e.stride;
    const double* Min = ((step & 1) ? e.Vs : e.V) + (int64_t)b * e.stride;
    double* Mout = ((step & 1) ? e.V : e.Vs) + (int64_t)b * e.stride;
    double* K = pb.K[b];
    auto in = [&](int i, int j) -> double { return first ? 0.5 * (A[i * ld + j] + A[j * ld + i]) : Min[i * ld + j]; };
    auto out = [&](int i, int j, double v) {
        if (last) K[(int64_t)i * n + j] = -v;
        else Mout[i * ld + j] = v;
    };
    // rows of the panel / of T in LDS
    auto crow = [&](int i) -> int { return BIG ? NB + (i - r0) : i; };   // a row of the slab
    auto trow = [&](int i) -> int { return BIG ? i - r0 : i; };
    const int piv0 = BIG ? 0 : kb;                                       // first pivot row

#ifdef SKF_PROBE_STAMPS
    long long ph[6] = {0, 0, 0, 0, 0, 0}, t_in = wall_clock64();
#define SKF_STAMP(i) { const long long now_ = wall_clock64(); ph[i] += now_ - t_in; t_in = now_; }
#else
#define SKF_STAMP(i)
#endif
    // ---- everything the step reads from memory is asked for up front: the panel (every row: the columns j of the update
    // come from it | BIG: pivot rows and slab rows), the diagonal of the input for the bounds of this step's pivots (as
    // sweep_inverse_kernel), and the wave's first tile of M -- its loads fly while wave 0 sweeps the pivot block
    auto panel_row = [&](int u) -> int {                                 // the matrix row behind LDS row ty + 16 u (-1: none)
        const int l = ty + 16 * u;
        if (!BIG) return l < n ? l : -1;
        const int i = l < NB ? kb + l : r0 + (l - NB);
        return (l < NB ? l < nb : i < r1) ? i : -1;
    };
    double pc[PCN];
#pragma unroll
    for (int u = 0; u < PCN; ++u) {
        const int i = panel_row(u);
        pc[u] = (i >= 0 && tx < nb) ? in(i, kb + tx) : 0.0;
    }
    double mx = 0.0;
    for (int k = tid; k < n; k += SWEEP_THREADS) mx = fmax(mx, fabs(A[k * ld + k]));
    const double akk = tid < nb ? A[(kb + tid) * ld + kb + tid] : 0.0;
    const int ncol = (n + 31) >> 5, nrow = (r1 - r0 + 31) >> 5;
    auto tile_at = [&](int blk, int& bi, int& bj) -> bool {              // false: pivot rows / columns only (or past the end)
        bi = r0 + (blk / ncol) * 32;
        bj = (blk % ncol) * 32;
        return blk < nrow * ncol && bi != kb && bj != kb;
    };
    struct Tile {
        MF::acc_t acc[2][2];
        double cb[BIG ? NB / MF::KT : 1][2];                             // BIG: the column operands C_j of the tile, all K steps
    };
    auto tile_load = [&](int bi, int bj, Tile& t) {                      // the accumulators start from M itself
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int i = bi + 16 * a + MF::d_row(lane, r);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = bj + 16 * q + MF::d_col(lane);
                    t.acc[a][q][r] = (i < n && j < n) ? in(i, j) : 0.0;
                }
            }
        if (BIG) {
#pragma unroll
            for (int ks = 0; ks < NB / MF::KT; ++ks) {
                const int kk = ks * MF::KT + MF::ab_k(lane);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = bj + 16 * q + MF::a_row(lane);
                    t.cb[BIG ? ks : 0][q] = (j < n && kk < nb) ? in(j, kb + kk) : 0.0;
                }
            }
        }
    };
    Tile t0;
    int bi0, bj0;
    const bool have0 = tile_at(wave, bi0, bj0);
    if (have0) tile_load(bi0, bj0, t0);
#pragma unroll
    for (int u = 0; u < PCN; ++u) {
        const int l = ty + 16 * u;
        if (BIG || l < n) Cs[l * LD + tx] = pc[u];
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    SKF_STAMP(0)
    if (wave == 0) {
        double smax = 0.0;
#pragma unroll
        for (int i = 0; i < SWEEP_THREADS / 64; ++i) smax = fmax(smax, red[i]);
        if (tid < nb) need[tid] = (akk > chol_diag_floor(n) * smax) ? fmax(rel_thr * akk, 0.0) : __builtin_inf();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const bool ok = sweep_pivot_block(Cs + piv0 * LD, Pv, rowk, need, nb, lane);
        if (lane == 0) s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    SKF_STAMP(1)
    if (!s_ok) {                                                         // (uniform, and the same verdict in every slab)
        if (tid == 0 && blockIdx.y == 0) e.chol_ok[b] = 0;
        return;
    }
    // ---- T = -(C Pv) for the rows of the slab
    {
        const int t0r = r0 >> 4, t1r = (r1 + 15) >> 4;
        for (int it = t0r + wave; it < t1r; it += SWEEP_THREADS / 64) {
            const int row = it * 16 + MF::a_row(lane);
            MF::acc_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < NB; k0 += MF::KT) {
                const int kk = k0 + MF::ab_k(lane);
                const double a = row < r1 ? Cs[crow(row) * LD + kk] : 0.0;
                a0 = MF::mma(a, Pv[kk * LD + MF::a_row(lane)], a0);
                a1 = MF::mma(a, Pv[kk * LD + 16 + MF::a_row(lane)], a1);
            }
#pragma unroll
            for (int r = 0; r < MF::NREG; ++r) {
                const int i = it * 16 + MF::d_row(lane, r);
                if (i < r1) {
                    Ts[trow(i) * LD + MF::d_col(lane)] = -a0[r];
                    Ts[trow(i) * LD + 16 + MF::d_col(lane)] = -a1[r];
                }
            }
        }
    }
    __syncthreads();
    SKF_STAMP(2)
    // ---- M_ij - T_i C_j^T outside the pivot rows / columns: a wave owns 32 x 32 outputs at a time
    {
        auto tile_finish = [&](int bi, int bj, Tile& t) {
            auto kstep = [&](int k0) {
                const int kk = k0 + MF::ab_k(lane);
                double ta[2], cb[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int i = bi + 16 * a + MF::a_row(lane);
                    ta[a] = i < r1 ? -Ts[trow(i) * LD + kk] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = bj + 16 * q + MF::a_row(lane);
                    if (BIG) cb[q] = t.cb[BIG ? k0 / MF::KT : 0][q];
                    else cb[q] = j < n ? Cs[j * LD + kk] : 0.0;
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < 2; ++q) t.acc[a][q] = MF::mma(ta[a], cb[q], t.acc[a][q]);
            };
            if (BIG) {                                                   // (unrolled: the operands of a K step are named registers)
#pragma unroll
                for (int k0 = 0; k0 < NB; k0 += MF::KT) kstep(k0);
            } else {
#pragma unroll 2
                for (int k0 = 0; k0 < NB; k0 += MF::KT) kstep(k0);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < MF::NREG; ++r) {
                    const int i = bi + 16 * a + MF::d_row(lane, r);
                    if (i >= r1 || (i >= kb && i < kb + nb)) continue;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int j = bj + 16 * q + MF::d_col(lane);
                        if (j < n && !(j >= kb && j < kb + nb)) out(i, j, t.acc[a][q][r]);
                    }
                }
        };
        if (have0) tile_finish(bi0, bj0, t0);
        for (int blk = wave + SWEEP_THREADS / 64; blk < nrow * ncol; blk += SWEEP_THREADS / 64) {
            int bi, bj;
            if (!tile_at(blk, bi, bj)) continue;
            Tile t;
            tile_load(bi, bj, t);
            tile_finish(bi, bj, t);
        }
    }
    SKF_STAMP(3)
    // ---- the pivot columns of the slab's rows <- T, the pivot rows at the slab's columns <- T^T, the pivot block <- -P
    for (int i = r0 + ty; i < r1; i += 16) {
        if (tx >= nb) continue;
        if (i >= kb && i < kb + nb) out(i, kb + tx, Pv[(i - kb) * LD + tx]);
        else out(i, kb + tx, Ts[trow(i) * LD + tx]);
    }
    for (int idx = tid; idx < nb * (r1 - r0); idx += SWEEP_THREADS) {
        const int c = idx / (r1 - r0), i = r0 + idx % (r1 - r0);
        if (!(i >= kb && i < kb + nb)) out(kb + c, i, Ts[trow(i) * LD + c]);
    }
    SKF_STAMP(4)
#ifdef SKF_PROBE_STAMPS
    if (tid == 0 && n >= 200 && blockIdx.y == 1 && (step == 0 || step == 3))
        printf("sweep_step n %d step %d: loads + panel %lld, pivot block sweep %lld, T %lld, update %lld, write-back %lld (x10 ns)\n", n, step,
               ph[0], ph[1], ph[2], ph[3], ph[4]);
#endif
    if (tid == 0 && blockIdx.y == 0) e.chol_ok[b] = last ? 1 : SWEEP_RUNNING;
}

static __global__ __launch_bounds__(256) void eigh_unpack_pinv_batched_kernel(PinvBatch pb, const double* __restrict__ VsAll,
                                                                       const double* __restrict__ VAll, int64_t stride,
                                                                       const int* __restrict__ chol_ok) {
    const int b = blockIdx.y;
    if (chol_ok[b] == 1) return;
    const int n = pb.c[b], n_pad = pb.n_pad[b];
    const double* Vs = VsAll + (int64_t)b * stride;
    const double* V = VAll + (int64_t)b * stride;
    double* K = pb.K[b];
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
        const int r = idx / n, c = idx % n;
        double s = 0.0;
        for (int k = 0; k < n_pad; ++k) s += Vs[r * n_pad + k] * V[c * n_pad + k];
        K[(int64_t)r * n + c] = s;
    }
}

}  // namespace skf
