// skf_small.h -- the DFMF iteration of a SMALL graph (every rank <= 64, a few thousand objects per type: the reference's
// own examples, BASELINE configs[0] / [1]) in three launches instead of ~33.
//
// On such graphs an iteration is a chain of dependent launches of a few microseconds each -- launch latency, not
// arithmetic (dicty: 0.30 ms for 80 MFLOP on the general schedule).  Here the chain is
//   1 small_contract_kernel   jobs: every P = R G_j row block (leaving its share of W = G_i^T P), every Q = R^T G_i share,
//                             every G^T G share, the constraint terms Theta-+ G of four rows.  The workgroup that finishes
//                             the LAST Gram share of a type sums the shares and inverts the matrix (sweep operator, the
//                             matrix in registers) while P / Q jobs are still in flight; the one that finishes the last P
//                             job of a relation sums W.                              (_dfmf.py:228-232, 254, 266, 284-292)
//                             A declined sweep (rank-deficient Gram matrix) falls back to deflation / eigen-solver there.
//   2 small_backbone_kernel   two workgroups per relation: S = K_i W K_j and the +- parts of S Gram_j S^T | S^T Gram_i S
//                                                                                                 (:236-239, 260-276)
//   3 small_update_kernel     per (type, 64 rows): the relation terms of E and D -- row sides, column sides, type term --
//                             in registers, the constraint terms added, G <- G sqrt(E / D) written in place  (:254-296)
// Same arithmetic as the general schedule (f64 c x c algebra, master-type n-sized products, nan_to_num where the
// reference has it); only the order of the sums differs.
#pragma once
#include "skf_kernels.h"

namespace skf {

constexpr int SM_MAXT = 16, SM_MAXR = 24, SM_MAXTH = 16;
constexpr int64_t SM_MAX_OBJECTS = 8192;     // objects per type up to which the three-launch schedule is chosen
constexpr int64_t SKF_THETA_SPARSE_DIV = 16; // a constraint with at most n * n / 16 non-zeros is compacted to CSR
struct SmType {
    void* G; void* E; void* D;
    double* Gram; double* K;
    int64_t n, gpart_off;        // offset (doubles) of this type's Gram shares in SmTables.gpart
    int c, n_gjobs;
    int has_theta, pad;          // sparse constraints on this type: their terms arrive in E / D (THETA jobs of launch 1)
};
struct SmRel {
    const void* R; int64_t ldr;
    void* P; void* Q;
    double* W; double* S; double* Bp; double* Bn; double* Dp; double* Dn;     // +- parts of S Gram_j S^T (row type) / S^T Gram_i S (column type)
    int64_t wpart_off;           // offset (doubles) of this relation's W shares in SmTables.wpart
    int row, col, n_pjobs, n_qparts;      // Q arrives as n_qparts shares (row ranges of the relation), summed where it is read
};
struct SmTheta {
    const int64_t* rp; const int* ci; const void* vv;
    int type, pad;
};
struct SmTables {
    int n_types, n_rels, n_thetas, nan_upd;
    double* wpart; double* gpart;
    int* tickets;                // [n_types + n_rels] finished Gram jobs per type / P jobs per relation (zero between launches)
    double* eigA; double* eigV; int* eigOk; int64_t eig_stride;
    double chol_thr;
    EighArgs eig;                // the fall-back of a declined sweep: deflation / eigen-solver scratch of the plan
    double defl_lo, defl_hi;
    int lds_rank, sweep_single;  // sweep_single: one pivot per barrier (SKF_SMALL_SWEEP4=0; tests: same bits either way)
    SmType t[SM_MAXT];
    SmRel r[SM_MAXR];
    SmTheta th[SM_MAXTH];
};
enum { SMJ_P = 0, SMJ_Q = 1, SMJ_GRAM = 2, SMJ_THETA = 3 };
// (Gram shares of 128 / 256 rows, measured on dicty: the share's K loop grows by what the sum of the shares saves -- 9.1 / 9.3 / 9.0 k it/s)
#ifndef SKF_SM_QROWS
#define SKF_SM_QROWS 256
#endif
constexpr int SM_QROWS = SKF_SM_QROWS, SM_GROWS = 64;       // rows of the relation per Q share / of the factor per Gram share
struct SmJob { int kind, idx, r0, nr, part, k0, nk, pad; };      // Q jobs: rows [k0, k0 + nk) of the relation, share `part`

// ---- matrix-core tile of a workgroup ---------------------------------------------------------------------------------
// 64 x 64 outputs per workgroup of 4 waves (2 x 2), K in tiles of 32 staged k-major in LDS (pitch 65); operands come
// from callables at(row, k) / at(k, col) that return 0 outside their matrix.  T = float: one v_mfma_f32_32x32x2_f32 tile
// per wave; T = double: 2 x 2 v_mfma_f64_16x16x4_f64 tiles per wave.
constexpr int SM_BK = 32, SM_LD = 65;      // (K tiles of 64 measured slower: 16 staged elements per thread and operand, their
                                            // index arithmetic, outweigh the saved round trips -- dicty 6940 -> 5850 it/s)
constexpr int SM_TILE_BYTES = 2 * SM_BK * SM_LD * 8;      // the two staging tiles of SmTile<double>: dynamic LDS of launches 1 and 4
template <typename T>
struct SmTile {
    typedef Mfma<T> MF;
    static constexpr int WR = 32 / MF::MT, WC = 32 / MF::NT;
    static constexpr int PER = 64 * SM_BK / 256;                 // elements of a K tile a thread stages, per operand
    typename MF::acc_t acc[WR][WC];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
            for (int j = 0; j < WC; ++j)
#pragma unroll
                for (int r = 0; r < MF::NREG; ++r) acc[i][j][r] = (T)0;
    }
    // acc += A * B over K: a_at(m, k), b_at(k, n); a_kfast: consecutive threads walk k (A rows contiguous in k), else m.
    // The operands of K tile t + 1 are fetched into registers before tile t goes through the matrix cores: the K loops of
    // a small graph are a handful of tiles, each one a full memory round trip if fetched on demand.
    template <class FA, class FB>
    __device__ __forceinline__ void mma(int K, FA a_at, FB b_at, bool a_kfast, T* As, T* Bs) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        T ra[PER], rb[PER];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int e = tid + 256 * q;
                const int m = a_kfast ? e / SM_BK : e % 64, kk = a_kfast ? e % SM_BK : e / 64;
                ra[q] = (k0 + kk < K) ? a_at(m, k0 + kk) : (T)0;
                const int kb = e / 64, n = e % 64;
                rb[q] = (k0 + kb < K) ? b_at(k0 + kb, n) : (T)0;
            }
        };
        if (K > 0) fetch(0);
        for (int k0 = 0; k0 < K; k0 += SM_BK) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int e = tid + 256 * q;
                const int m = a_kfast ? e / SM_BK : e % 64, kk = a_kfast ? e % SM_BK : e / 64;
                As[kk * SM_LD + m] = ra[q];
                Bs[(e / 64) * SM_LD + e % 64] = rb[q];
            }
            __syncthreads();
            if (k0 + SM_BK < K) fetch(k0 + SM_BK);
#pragma unroll
            for (int kk = 0; kk < SM_BK; kk += MF::KT) {
                T av[WR], bv[WC];
                const int kr = kk + MF::ab_k(lane);
#pragma unroll
                for (int i = 0; i < WR; ++i) av[i] = As[kr * SM_LD + wm0 + i * MF::MT + MF::a_row(lane)];
#pragma unroll
                for (int j = 0; j < WC; ++j) bv[j] = Bs[kr * SM_LD + wn0 + j * MF::NT + MF::a_row(lane)];
#pragma unroll
                for (int i = 0; i < WR; ++i)
#pragma unroll
                    for (int j = 0; j < WC; ++j) acc[i][j] = MF::mma(av[i], bv[j], acc[i][j]);
            }
        }
    }
    // f(row, col, value) for every accumulator element of this lane
    template <class F>
    __device__ __forceinline__ void for_each(F f) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
            for (int j = 0; j < WC; ++j)
#pragma unroll
                for (int r = 0; r < MF::NREG; ++r)
                    {
                    const T val = acc[i][j][r];
                    f(wm0 + i * MF::MT + MF::d_row(lane, r), wn0 + j * MF::NT + MF::d_col(lane), val);
                }
    }
};

// t1 += A * B1 and t2 += A * B2 over the same K tiles of A (the two type-term products G sum B- / G sum B+ of the update
// launch): one staged A tile, one round trip per K tile for both -- every K tile whose operand another workgroup wrote in
// the previous launch costs a trip through memory (~2 us on dicty).  Same order of the K tiles, separate accumulators: the
// bits of two mma() calls.
constexpr int SM_TILE3_BYTES = 3 * SM_BK * SM_LD * 8 + 2 * 64 * 64 * 8;      // + the two c x c sums of the type term (update launch)
template <typename T, class FA, class FB1, class FB2>
__device__ __forceinline__ void sm_mma_dual(SmTile<T>& t1, SmTile<T>& t2, int K, FA a_at, FB1 b1_at, FB2 b2_at, bool a_kfast,
                                            T* As, T* Bs1, T* Bs2) {
    typedef Mfma<T> MF;
    constexpr int PER = SmTile<T>::PER, WR = SmTile<T>::WR, WC = SmTile<T>::WC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    T ra[PER], rb1[PER], rb2[PER];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q;
            const int m = a_kfast ? e / SM_BK : e % 64, kk = a_kfast ? e % SM_BK : e / 64;
            ra[q] = (k0 + kk < K) ? a_at(m, k0 + kk) : (T)0;
            const int kb = e / 64, n = e % 64;
            rb1[q] = (k0 + kb < K) ? b1_at(k0 + kb, n) : (T)0;
            rb2[q] = (k0 + kb < K) ? b2_at(k0 + kb, n) : (T)0;
        }
    };
    if (K > 0) fetch(0);
    for (int k0 = 0; k0 < K; k0 += SM_BK) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q;
            const int m = a_kfast ? e / SM_BK : e % 64, kk = a_kfast ? e % SM_BK : e / 64;
            As[kk * SM_LD + m] = ra[q];
            Bs1[(e / 64) * SM_LD + e % 64] = rb1[q];
            Bs2[(e / 64) * SM_LD + e % 64] = rb2[q];
        }
        __syncthreads();
        if (k0 + SM_BK < K) fetch(k0 + SM_BK);
#pragma unroll
        for (int kk = 0; kk < SM_BK; kk += MF::KT) {
            T av[WR], bv1[WC], bv2[WC];
            const int kr = kk + MF::ab_k(lane);
#pragma unroll
            for (int i = 0; i < WR; ++i) av[i] = As[kr * SM_LD + wm0 + i * MF::MT + MF::a_row(lane)];
#pragma unroll
            for (int j = 0; j < WC; ++j) {
                bv1[j] = Bs1[kr * SM_LD + wn0 + j * MF::NT + MF::a_row(lane)];
                bv2[j] = Bs2[kr * SM_LD + wn0 + j * MF::NT + MF::a_row(lane)];
            }
#pragma unroll
            for (int i = 0; i < WR; ++i)
#pragma unroll
                for (int j = 0; j < WC; ++j) {
                    t1.acc[i][j] = MF::mma(av[i], bv1[j], t1.acc[i][j]);
                    t2.acc[i][j] = MF::mma(av[i], bv2[j], t2.acc[i][j]);
                }
        }
    }
}

// What the sweep declined: rank-revealing deflation, then the Jacobi eigen-solver for what that declines too, then
// K = Vs V^T -- the three fall-back launches of the general schedule (pchol_pinv_kernel, jacobi_eigh_kernel,
// eigh_unpack_pinv_batched_kernel) run by the workgroup that holds the matrix, inside launch 1.  Rare (a rank-deficient
// Gram matrix) and slow (milliseconds): the rest of the launch does not wait for it, the next launch does.
__device__ __forceinline__ void small_fallback_body(const SmTables* __restrict__ tb, const int b) {
    const EighArgs& e = tb->eig;
    const SmType& ty = tb->t[b];
    const int n = ty.c, n_pad = e.n[b];
    {   // the packed, zero-padded copy of the Gram matrix the fall-back kernels work on
        double* A = e.A + (int64_t)b * e.stride;
        for (int idx = threadIdx.x; idx < n_pad * n_pad; idx += blockDim.x) {
            const int r = idx / n_pad, c = idx % n_pad;
            A[idx] = (r < n && c < n) ? ty.Gram[r * n + c] : 0.0;
        }
        __syncthreads();
    }
    pchol_pinv_body(e, b, tb->defl_lo, tb->defl_hi, tb->lds_rank);
    __syncthreads();
    jacobi_eigh_body(e, b);
    __syncthreads();
    const double* Vs = e.Vs + (int64_t)b * e.stride;
    const double* V = e.V + (int64_t)b * e.stride;
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const int r = idx / n, c = idx % n;
        double s2 = 0.0;
        for (int k = 0; k < n_pad; ++k) s2 += Vs[r * n_pad + k] * V[c * n_pad + k];
        ty.K[(int64_t)r * n + c] = s2;
    }
}

// ---- 1b -----------------------------------------------------------------------------------------------------------
// p[0] + p[stride] + ... (count terms) per element in a fixed order: four chains (share q on chain q mod 4, the tail on chain
// 0), (s0 + s1) + (s2 + s3) -- for EL elements at once, element el at p + off[el] (off[el] < 0: none): the loads of 8 shares of
// all EL elements are in flight together (a thread of the summing workgroup owns up to 16 elements).
template <int EL>
__device__ __forceinline__ void sum_shares_multi(const double* __restrict__ p, const int (&off)[EL], int64_t stride, int count,
                                                 double (&out)[EL]) {
    double s[EL][4];
#pragma unroll
    for (int el = 0; el < EL; ++el)
#pragma unroll
        for (int c = 0; c < 4; ++c) s[el][c] = 0.0;
    const int full = count & ~3;                    // shares q < full go to chain q mod 4, the tail to chain 0
    for (int q0 = 0; q0 < count; q0 += 8) {
        double v[EL][8];
#pragma unroll
        for (int el = 0; el < EL; ++el)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[el][u] = (off[el] >= 0 && q0 + u < count) ? p[off[el] + (int64_t)(q0 + u) * stride] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (q0 + u < full) {                    // (uniform)
#pragma unroll
                for (int el = 0; el < EL; ++el) s[el][u & 3] += v[el][u];
            } else if (q0 + u < count) {
#pragma unroll
                for (int el = 0; el < EL; ++el) s[el][0] += v[el][u];
            }
        }
    }
#pragma unroll
    for (int el = 0; el < EL; ++el) out[el] = (s[el][0] + s[el][1]) + (s[el][2] + s[el][3]);
}

// One workgroup per object type -- the one that finishes the type's LAST Gram job in launch 1, so that the inverse runs
// underneath the P and Q jobs still in flight: Gram = nan_to_num(sum of its shares) (_dfmf.py:229) and K = Gram^-1 by
// the sweep operator -- n steps of a rank-one update of the whole matrix.  The matrix lives in
// REGISTERS: thread (ti, tj) of the 16 x 16 grid owns the 4 x 4 block at (4 ti, 4 tj); per step the only LDS traffic is
// the pivot column (published by its 16 owners into a double buffer, read back as two 32-byte vectors per thread) and
// there is ONE barrier.  (A one-wave Cholesky whose every inner product is a chain of LDS round trips: 91 us at order
// 50; the sweep with the matrix in LDS, 12 LDS reads per thread and step: 52 us; this form: see profiles/.)
// The pivot of step k is the Schur complement the Cholesky factorisation would take the root of: same verdict, same
// thresholds as chol_inverse_small_kernel; a failed pivot leaves the matrix to small_fallback_body.
constexpr int SM_PINV_THREADS = 256;
// (M: 64 x SM_LD doubles of LDS, free to clobber -- the staging tiles of the calling workgroup)
__device__ __forceinline__ void small_pinv_body(const SmTables* __restrict__ tb, const int t, double* M) {
    __shared__ __attribute__((aligned(32))) double col[2][4][64];
    __shared__ double diag0[64], need[64];
    const int tid = threadIdx.x;
    const SmType& ty = tb->t[t];
    const int n = ty.c;
#ifdef SKF_PROBE_STAMPS                    // probe builds only (tools/build_probe_libs.sh stamps:-DSKF_PROBE_STAMPS): where launch 1 spends its time
    long long stamp[8];
    stamp[0] = wall_clock64();
#endif
    for (int e = tid; e < 64 * SM_LD; e += SM_PINV_THREADS) M[e] = 0.0;
    __syncthreads();
    {
        const double* part = tb->gpart + ty.gpart_off;
        // Only the lower triangle is summed: a share is G^T G of its rows on the matrix cores -- element (a, b) and element
        // (b, a) are the same products added in the same order, the same bits -- and the summing workgroup's time is the
        // traffic of the shares (400 KB from other XCDs for dicty's genes: 14 us of the launch's 48), not their latency.
        const int ntri = n * (n + 1) / 2;
        for (int t0 = tid; t0 < ntri; t0 += 8 * SM_PINV_THREADS) {
            double sums[8];
            int off[8], ri[8], rj[8];
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const int t = t0 + el * SM_PINV_THREADS;
                int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                while (i * (i + 1) / 2 > t) --i;
                ri[el] = i;
                rj[el] = t - i * (i + 1) / 2;
                off[el] = t < ntri ? i * n + rj[el] : -1;
            }
            sum_shares_multi<8>(part, off, n * n, ty.n_gjobs, sums);
#pragma unroll
            for (int el = 0; el < 8; ++el)
                if (off[el] >= 0) {
                    const double s = nan_to_num(sums[el]);
                    ty.Gram[ri[el] * n + rj[el]] = s;
                    ty.Gram[rj[el] * n + ri[el]] = s;
                    M[ri[el] * SM_LD + rj[el]] = s;
                    M[rj[el] * SM_LD + ri[el]] = s;
                }
        }
    }
    __syncthreads();
#ifdef SKF_PROBE_STAMPS
    stamp[1] = wall_clock64();
#endif
    const int ti = tid >> 4, tj = tid & 15, i0 = 4 * ti, j0 = 4 * tj;
    double m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) m[a][b] = 0.5 * (M[(i0 + a) * SM_LD + j0 + b] + M[(j0 + b) * SM_LD + i0 + a]);
    if (tid < 64) diag0[tid] = 0.5 * (M[tid * SM_LD + tid] + M[tid * SM_LD + tid]);
    // column k of the current matrix into col[buf][k & 3] (= row k: the matrix stays symmetric).  Its owners are the 16
    // threads of block column k / 4; they publish all four of their columns, so that no register is indexed by k (a
    // select over m[a][k & 3] sends the whole block to scratch memory: 105 us instead of 52).
    auto publish = [&](int k, int buf) {
        if (tj != (k >> 2)) return;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) col[buf][b][i0 + a] = m[a][b];
    };
    publish(0, 0);
    __syncthreads();
    double mx = 0.0;
    for (int i = 0; i < n; ++i) mx = fmax(mx, fabs(diag0[i]));
    const double floor_ = chol_diag_floor(n) * mx, thr = tb->chol_thr;
    // the three tests on pivot k -- a_kk > floor, pivot > thr a_kk, pivot > 0 -- as ONE bound the pivot has to exceed
    // (+inf where the diagonal entry fails, NaN included), so that a step is one LDS round trip before its arithmetic
    if (tid < 64) {
        const double akk = diag0[tid];
        need[tid] = (akk > floor_) ? fmax(thr * akk, 0.0) : __builtin_inf();
    }
    __syncthreads();
#ifdef SKF_PROBE_STAMPS
    stamp[2] = wall_clock64();
#endif
    int ok = 1;
    // Four pivots per barrier (round 4).  A step of the loop below is one LDS round trip and one barrier for 16 fused
    // multiply-adds per thread: latency, 0.7 us per pivot.  The owners of block column kb publish its four columns ONCE,
    // as they stand before pivot 4 kb; every thread then replays the four sweeps on what it needs of them -- the panel
    // rows of its own rows (U), of its own columns taken as rows (V: the role c[j0 + b] plays below) and of the pivot
    // block itself (Wp, the same in every thread) -- with exactly the operations their owners would have applied between
    // two barriers: x <- fma(-c_r, c_piv d, x), row k -> c_piv d.  Same inputs, same instructions: the same bits as the
    // one-pivot loop (which stays behind SKF_SMALL_SWEEP4=0 and is compared with this one in the tests).
    if (!tb->sweep_single) {
        for (int k0 = 0; k0 < n && ok; k0 += 4) {
            typedef double vec4 __attribute__((ext_vector_type(4)));
            const int buf = (k0 >> 2) & 1;
            double U[4][4], V[4][4], Wp[4][4];             // [row][q]: entry (row, k0 + q) of the current matrix
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const vec4 u = *(const vec4*)(&col[buf][q][i0]), v = *(const vec4*)(&col[buf][q][j0]), w = *(const vec4*)(&col[buf][q][k0]);
#pragma unroll
                for (int a = 0; a < 4; ++a) { U[a][q] = u[a]; V[a][q] = v[a]; Wp[a][q] = w[a]; }
            }
            const int nq = n - k0 < 4 ? n - k0 : 4;
            const bool row_hit = (i0 == k0), col_hit = (j0 == k0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nq && ok) {                                     // (uniform)
                    const int k = k0 + q;
                    const double piv = Wp[q][q], bound = need[k];
                    if (!(piv > bound)) {
                        ok = 0;
                    } else {
                        // x <- fma(-c_i, c_j d, x) everywhere, then the swept row and column written over it with what the
                        // one-pivot step's selects produce there: row k  fma(1, c_j d, 0) = c_j d,  column k
                        // fma(-c_i, -d, 0) = c_i d (one rounding either way),  (k, k)  -d.  In a block of four pivots row
                        // k is row q of the threads with ti == kb, column k column q of those with tj == kb: a compile-time
                        // register index under a branch instead of two selects per element.
                        const double d = 1.0 / piv;
                        double cjd[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) cjd[b] = V[b][q] * d;
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 4; ++b) m[a][b] = fma(-U[a][q], cjd[b], m[a][b]);
                        if (row_hit) {
#pragma unroll
                            for (int b = 0; b < 4; ++b) m[q][b] = cjd[b];
                        }
                        if (col_hit) {
#pragma unroll
                            for (int a = 0; a < 4; ++a) m[a][q] = U[a][q] * d;
                            if (row_hit) m[q][q] = -d;
                        }
                        // the columns still to come in this block, as their owners would have updated them
#pragma unroll
                        for (int q2 = q + 1; q2 < 4; ++q2) {
                            const double pd = Wp[q2][q] * d;            // c[k0 + q2] d: column k0 + q2 is not column k
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                U[a][q2] = fma(-U[a][q], pd, U[a][q2]);
                                V[a][q2] = fma(-V[a][q], pd, V[a][q2]);
                                Wp[a][q2] = fma(-Wp[a][q], pd, Wp[a][q2]);
                            }
                            if (row_hit) U[q][q2] = pd;                 // row k of the panel: fma(1, c d, 0)
                            if (col_hit) V[q][q2] = pd;
                            Wp[q][q2] = pd;
                        }
                    }
                }
            }
            if (ok && k0 + 4 < n) publish(k0 + 4, buf ^ 1);
            __syncthreads();
        }
    } else
    for (int k = 0; k < n; ++k) {
        const double* c = col[k & 1][k & 3];
        typedef double vec4 __attribute__((ext_vector_type(4)));
        const vec4 cr = *(const vec4*)(c + i0), cc4 = *(const vec4*)(c + j0);
        const double piv = c[k], bound = need[k];
        if (!(piv > bound)) {                                      // (uniform: every thread reads the same words)
            ok = 0;
            break;
        }
        const double d = 1.0 / piv;
        // swept matrix: m_ij - c_i c_j d in general; c_j d in row k, c_i d in column k, -d at (k, k) -- as ONE fused
        // multiply-add per element with c_i -> -1 in row k, c_j d -> -d in column k and the old value dropped there
        double ci[4], cjd[4];
        bool rk[4], ck[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            rk[a] = (i0 + a == k);
            ck[a] = (j0 + a == k);
            ci[a] = rk[a] ? -1.0 : cr[a];
            cjd[a] = ck[a] ? -d : cc4[a] * d;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double old = (rk[a] || ck[b]) ? 0.0 : m[a][b];
                m[a][b] = fma(-ci[a], cjd[b], old);
            }
        if (k + 1 < n) publish(k + 1, (k + 1) & 1);
        __syncthreads();
    }
#ifdef SKF_PROBE_STAMPS
    stamp[3] = wall_clock64();
    if (tid == 0 && n >= 32)
        printf("pinv type %d n %d: sum of shares %lld, set-up %lld, sweep %lld (x10 ns), entered at %lld\n", t, n, stamp[1] - stamp[0],
               stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[0]);
#endif
    if (ok) {                                                      // all pivots swept: m = -Gram^-1
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (i0 + a < n && j0 + b < n) ty.K[(i0 + a) * n + j0 + b] = -m[a][b];
    }
    if (tid == 0) tb->eigOk[t] = ok;
    if (!ok) {                                                     // (uniform)
        __syncthreads();                                           // the verdict is in memory before the bodies read it
        small_fallback_body(tb, t);
    }
}

// true in every thread of the LAST workgroup to get here out of `total` (the others' global writes are visible to it);
// the counter is back at zero for the next iteration
__device__ __forceinline__ bool last_arrival(int* counter, int total) {
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tk = atomicAdd(counter, 1);
        s_last = (tk == total - 1) ? 1 : 0;
        if (s_last) *counter = 0;
    }
    __syncthreads();
    const bool last = s_last != 0;
    if (last) __threadfence();
    return last;
}

// ---- 1 ------------------------------------------------------------------------------------------------------------
// BATCH: blockIdx.y = the plan of a batch of restarts (skf_iterate_batch), its tables looked up in `tbs`.  A single plan
// passes its tables as the kernel argument itself: behind the look-up the compiler no longer treats the table entries as
// invariant kernel inputs, and the three launches of an iteration took 155 us instead of 132 (dicty, profiles/r03_dicty_batch_ab.txt).
template <typename T, bool BATCH>
__global__ __launch_bounds__(256) void small_contract_kernel(const SmTables* __restrict__ tb0, const SmTables* const* __restrict__ tbs,
                                                             const SmJob* __restrict__ jobs) {
    const SmTables* __restrict__ tb = BATCH ? tbs[blockIdx.y] : tb0;
    HIP_DYNAMIC_SHARED(double, sm_tiles)     // 2 x SM_BK x SM_LD = 64 x SM_LD doubles: the two staging tiles, or the matrix of the sweep
    double* As = sm_tiles;
    double* Bs = sm_tiles + SM_BK * SM_LD;
    const SmJob jb = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    if (jb.kind == SMJ_GRAM) {           // share of G^T G from rows r0 .., f64 accumulation
#ifdef SKF_PROBE_STAMPS
        const long long g0 = wall_clock64();
#endif
        const SmType& t = tb->t[jb.idx];
        const T* G = (const T*)t.G;
        const int c = t.c;
        SmTile<double> w;
        w.zero();
        auto g_at = [&](int a, int k) -> double { return (a < c && k < jb.nr) ? (double)G[(int64_t)(jb.r0 + k) * c + a] : 0.0; };
        w.mma(jb.nr, g_at, [&](int k, int b) { return g_at(b, k); }, false, As, Bs);
        double* out = tb->gpart + t.gpart_off + (int64_t)jb.part * c * c;
        w.for_each([&](int a, int b, double v) { if (a < c && b < c) out[a * c + b] = v; });
#ifdef SKF_PROBE_STAMPS
        const long long g1 = wall_clock64();
        const bool last_one = last_arrival(tb->tickets + jb.idx, t.n_gjobs);
        if (last_one && tid == 0 && c >= 32) printf("gram job block %d: started %lld, product + store %lld, arrival %lld (x10 ns)\n", (int)blockIdx.x, g0, g1 - g0, wall_clock64() - g1);
        if (last_one) small_pinv_body(tb, jb.idx, sm_tiles);
        if (last_one && tid == 0 && c >= 32) printf("   done at %lld\n", wall_clock64());
        return;
#endif
        if (last_arrival(tb->tickets + jb.idx, t.n_gjobs)) small_pinv_body(tb, jb.idx, sm_tiles);
        return;
    }
    if (jb.kind == SMJ_THETA) {          // rows r0 .. of a constrained type, one wave per row: E = Theta- G, D = Theta+ G
        const SmType& t = tb->t[jb.idx]; //                                                                 (_dfmf.py:284-292)
        const int lane = tid & 63, c = t.c;
        for (int rr = tid >> 6; rr < jb.nr; rr += 4) {
            const int64_t row = jb.r0 + rr;
            T e = (T)0, d = (T)0;
            for (int k = 0; k < tb->n_thetas; ++k) {
                const SmTheta& th = tb->th[k];
                if (th.type != jb.idx) continue;
                theta_row_walk<T>(th.ci, (const T*)th.vv, th.rp[row], th.rp[row + 1], (const T*)t.G, c, lane, e, d);
            }
            if (lane < c) {
                ((T*)t.E)[row * c + lane] = e;
                ((T*)t.D)[row * c + lane] = d;
            }
        }
        return;
    }
    const SmRel& r = tb->r[jb.idx];
    const SmType& ti = tb->t[r.row];
    const SmType& tj = tb->t[r.col];
    const T* R = (const T*)r.R;
    const int ci = ti.c, cj = tj.c;
    SmTile<T> acc;
    acc.zero();
    if (jb.kind == SMJ_P) {              // P[r0 + m][:] = sum_k R[r0 + m][k] G_j[k][:]
        const T* Gj = (const T*)tj.G;
        const T* Gi = (const T*)ti.G;
        const int nj = (int)tj.n;
        acc.mma(nj, [&](int m, int k) { return m < jb.nr ? R[(int64_t)(jb.r0 + m) * r.ldr + k] : (T)0; },
                [&](int k, int n) { return n < cj ? Gj[(int64_t)k * cj + n] : (T)0; }, true, (T*)As, (T*)Bs);
        T* P = (T*)r.P;
        acc.for_each([&](int m, int n, T v) { if (m < jb.nr && n < cj) P[(int64_t)(jb.r0 + m) * cj + n] = v; });
        __syncthreads();                 // (the rows of P this workgroup just wrote are read back below)
        // share of W = G_i^T P from these rows, f64 accumulation (as the general schedule's G^T P product)
        SmTile<double> w;
        w.zero();
        w.mma(jb.nr, [&](int a, int k) -> double { return a < ci ? (double)Gi[(int64_t)(jb.r0 + k) * ci + a] : 0.0; },
              [&](int k, int b) -> double { return b < cj ? (double)P[(int64_t)(jb.r0 + k) * cj + b] : 0.0; }, false, As, Bs);
        double* out = tb->wpart + r.wpart_off + (int64_t)jb.part * ci * cj;
        w.for_each([&](int a, int b, double v) { if (a < ci && b < cj) out[a * cj + b] = v; });
        if (last_arrival(tb->tickets + tb->n_types + jb.idx, r.n_pjobs)) {      // W = the sum of its shares, fixed order
            const double* part = tb->wpart + r.wpart_off;
            for (int e0 = tid; e0 < ci * cj; e0 += 8 * 256) {
                double sums[8];
                int off[8];
#pragma unroll
                for (int el = 0; el < 8; ++el) off[el] = e0 + el * 256 < ci * cj ? e0 + el * 256 : -1;
                sum_shares_multi<8>(part, off, ci * cj, r.n_pjobs, sums);
#pragma unroll
                for (int el = 0; el < 8; ++el)
                    if (off[el] >= 0) r.W[off[el]] = sums[el];
            }
        }
    } else {                             // share `part` of Q[c0 + m][:] = sum over the rows k0 .. k0 + nk of R[k][c0 + m] G_i[k][:]
        const T* Gi = (const T*)ti.G;
        acc.mma(jb.nk, [&](int m, int k) { return m < jb.nr ? R[(int64_t)(jb.k0 + k) * r.ldr + jb.r0 + m] : (T)0; },
                [&](int k, int n) { return n < ci ? Gi[(int64_t)(jb.k0 + k) * ci + n] : (T)0; }, false, (T*)As, (T*)Bs);
        T* Q = (T*)r.Q + (int64_t)jb.part * tj.n * ci;       // r.Q: [n_qparts][n_j][c_i]
        acc.for_each([&](int m, int n, T v) { if (m < jb.nr && n < ci) Q[(int64_t)(jb.r0 + m) * ci + n] = v; });
    }
}

// ---- 6 ------------------------------------------------------------------------------------------------------------
// TWO workgroups per relation, each a chain of four c x c x c products on the f64 matrix cores with operands and
// intermediates in LDS: both form S = K_i W K_j (the same instructions on the same data: bit-identical), the even one goes
// on to the +- parts of S Gram_j S^T (row type) and writes S, the odd one to those of S^T Gram_i S (column type) -- the two
// branches are independent, so the dependent chain is four products deep instead of six.
// (dynamic LDS: 2 x 64 x 65 doubles + the two staging tiles of SmTile)
template <bool BATCH>
__global__ __launch_bounds__(256) void small_backbone_kernel(const SmTables* __restrict__ tb0, const SmTables* const* __restrict__ tbs) {
    const SmTables* __restrict__ tb = BATCH ? tbs[blockIdx.y] : tb0;
    HIP_DYNAMIC_SHARED(double, sm)
    const SmRel& r = tb->r[blockIdx.x >> 1];
    const bool col_side = (blockIdx.x & 1) != 0;
    const SmType& ti = tb->t[r.row];
    const SmType& tj = tb->t[r.col];
    const int ci = ti.c, cj = tj.c, tid = threadIdx.x;
    double* X = sm;                       // T1 = K_i W, then U
    double* S = X + 64 * SM_LD;           // S
    double* As = S + 64 * SM_LD;
    double* Bs = As + SM_BK * SM_LD;
    SmTile<double> t;
    // T1 = K_i W
    t.zero();
    t.mma(ci, [&](int a, int k) { return a < ci ? ti.K[a * ci + k] : 0.0; }, [&](int k, int b) { return b < cj ? r.W[k * cj + b] : 0.0; },
          true, As, Bs);
    t.for_each([&](int a, int b, double v) { X[a * SM_LD + b] = v; });
    __syncthreads();                      // (the next product fetches its first operands before its own first barrier)
    // S = nan_to_num(T1 K_j)                                                                          (_dfmf.py:236-239)
    t.zero();
    t.mma(cj, [&](int a, int k) { return X[a * SM_LD + k]; }, [&](int k, int b) { return b < cj ? tj.K[k * cj + b] : 0.0; }, true, As, Bs);
    t.for_each([&](int a, int b, double v) {
        v = nan_to_num(v);
        S[a * SM_LD + b] = (a < ci && b < cj) ? v : 0.0;
        if (!col_side && a < ci && b < cj) r.S[a * cj + b] = v;
    });
    __syncthreads();
    if (!col_side) {
        // U = S Gram_j ; B = U S^T, split                                                             (_dfmf.py:260-264)
        t.zero();
        t.mma(cj, [&](int a, int k) { return S[a * SM_LD + k]; }, [&](int k, int b) { return b < cj ? tj.Gram[k * cj + b] : 0.0; }, true, As, Bs);
        t.for_each([&](int a, int b, double v) { X[a * SM_LD + b] = v; });
        __syncthreads();
        t.zero();
        t.mma(cj, [&](int a, int k) { return X[a * SM_LD + k]; }, [&](int k, int c2) { return S[c2 * SM_LD + k]; }, true, As, Bs);
        t.for_each([&](int a, int c2, double v) {
            if (a < ci && c2 < ci) {
                if (tb->nan_upd) v = nan_to_num(v);
                r.Bp[a * ci + c2] = v > 0.0 ? v : 0.0;
                r.Bn[a * ci + c2] = v > 0.0 ? 0.0 : -v;
            }
        });
        return;
    }
    // U = Gram_i S ; D = S^T U, split                                                                 (_dfmf.py:272-276)
    t.zero();
    t.mma(ci, [&](int a, int k) { return a < ci ? ti.Gram[a * ci + k] : 0.0; }, [&](int k, int b) { return S[k * SM_LD + b]; }, true, As, Bs);
    t.for_each([&](int a, int b, double v) { X[a * SM_LD + b] = v; });
    __syncthreads();
    t.zero();
    t.mma(ci, [&](int a, int k) { return S[k * SM_LD + a]; }, [&](int k, int c2) { return X[k * SM_LD + c2]; }, false, As, Bs);
    t.for_each([&](int a, int c2, double v) {
        if (a < cj && c2 < cj) {
            if (tb->nan_upd) v = nan_to_num(v);
            r.Dp[a * cj + c2] = v > 0.0 ? v : 0.0;
            r.Dn[a * cj + c2] = v > 0.0 ? 0.0 : -v;
        }
    });
}

// ---- 7 ------------------------------------------------------------------------------------------------------------
// job = (type jb.idx, rows r0 .. r0 + nr <= 64): every term of E and D of those rows on the matrix cores, the +- split of a
// relation side in registers (nan_to_num first, as the reference has it), the type term accumulated straight into E / D
template <typename T, bool BATCH>
__global__ __launch_bounds__(256) void small_update_kernel(const SmTables* __restrict__ tb0, const SmTables* const* __restrict__ tbs,
                                                           const SmJob* __restrict__ jobs) {
    const SmTables* __restrict__ tb = BATCH ? tbs[blockIdx.y] : tb0;
    HIP_DYNAMIC_SHARED(double, sm_tiles)
    T* As = (T*)sm_tiles;
    T* Bs = (T*)(sm_tiles + SM_BK * SM_LD);
    const SmJob jb = jobs[blockIdx.x];
    const SmType& ty = tb->t[jb.idx];
    const int c = ty.c, tid = threadIdx.x;
#ifdef SKF_PROBE_STAMPS
    const long long u0 = wall_clock64();
#endif
    SmTile<T> e, d, v;
    e.zero();
    d.zero();
    auto split_add = [&]() {
#pragma unroll
        for (int i = 0; i < SmTile<T>::WR; ++i)
#pragma unroll
            for (int j = 0; j < SmTile<T>::WC; ++j)
#pragma unroll
                for (int q = 0; q < Mfma<T>::NREG; ++q) {
                    T x = v.acc[i][j][q];
                    if (tb->nan_upd) x = nan_to_num(x);
                    e.acc[i][j][q] += x > (T)0 ? x : (T)0;
                    d.acc[i][j][q] += x > (T)0 ? (T)0 : -x;
                }
    };
    for (int k = 0; k < tb->n_rels; ++k) {
        const SmRel& r = tb->r[k];
        const int ci = tb->t[r.row].c, cj = tb->t[r.col].c;
        if (r.row == jb.idx) {                     // (P S^T)+-                                      (_dfmf.py:254-258)
            const T* P = (const T*)r.P;
            v.zero();
            v.mma(cj, [&](int m, int b) { return m < jb.nr ? P[(int64_t)(jb.r0 + m) * cj + b] : (T)0; },
                  [&](int b, int a) { return a < ci ? (T)r.S[a * cj + b] : (T)0; }, true, As, Bs);
            split_add();
        }
        if (r.col == jb.idx) {                     // (Q S)+-                                        (_dfmf.py:266-270)
            const T* Q = (const T*)r.Q;
            const int64_t qs = tb->t[r.col].n * ci;       // Q arrives as n_qparts shares
            v.zero();
            const int nqp = r.n_qparts;
            v.mma(ci, [&](int m, int a) {        // the shares are loaded eight at a time, then added in their order: one round
                      T x = (T)0;                //  trip per element instead of one per share (25 -> 10 us for a job of dicty's)
                      if (m < jb.nr) {
                          const T* q0 = Q + (int64_t)(jb.r0 + m) * ci + a;
                          for (int z0 = 0; z0 < nqp; z0 += 8) {
                              T sh[8];
#pragma unroll
                              for (int u = 0; u < 8; ++u) sh[u] = (z0 + u < nqp) ? q0[(z0 + u) * qs] : (T)0;
#pragma unroll
                              for (int u = 0; u < 8; ++u)
                                  if (z0 + u < nqp) x += sh[u];
                          }
                      }
                      return x;
                  },
                  [&](int a, int b) { return b < cj ? (T)r.S[a * cj + b] : (T)0; }, true, As, Bs);
            split_add();
        }
    }
#ifdef SKF_PROBE_STAMPS
    const long long u1 = wall_clock64();
#endif
    // type term: E += G sum B-, D += G sum B+ (sums over the relations of the type, rounded to T once; _dfmf.py:278-282)
    // The sums are staged in LDS once per workgroup (c x c each, behind the three staging tiles): inside the K loop every
    // element was a walk over the relation table with two dependent loads per relation -- 22 us of the 30 a job of dicty's
    // genes took (time stamps of a probe build).
    const T* G = (const T*)ty.G;
    auto g_at = [&](int m, int k) { return m < jb.nr ? G[(int64_t)(jb.r0 + m) * c + k] : (T)0; };
    T* Bneg = (T*)(sm_tiles + 3 * SM_BK * SM_LD);
    T* Bpos = Bneg + SMALLC * SMALLC;
    for (int el = tid; el < c * c; el += 256) {
        double sn = 0.0, sp = 0.0;
        for (int q = 0; q < tb->n_rels; ++q) {
            const SmRel& r = tb->r[q];
            if (r.row == jb.idx) { sn += r.Bn[el]; sp += r.Bp[el]; }
            if (r.col == jb.idx) { sn += r.Dn[el]; sp += r.Dp[el]; }
        }
        Bneg[el] = (T)sn;
        Bpos[el] = (T)sp;
    }
    __syncthreads();
    sm_mma_dual<T>(e, d, c, g_at, [&](int k, int a) { return a < c ? Bneg[k * c + a] : (T)0; },
                   [&](int k, int a) { return a < c ? Bpos[k * c + a] : (T)0; }, true, As, Bs, (T*)(sm_tiles + 2 * SM_BK * SM_LD));
    // G <- G * sqrt(E / max(D, eps)) for the rows of this job (_dfmf.py:294-296; the arithmetic of mult_update_kernel): E and D
    // never leave the registers; the constraint terms were left in the E / D arrays by the THETA jobs of the first launch.
    // (Every read of these rows of G -- the type term above -- is behind the last barrier of the product.)
#ifdef SKF_PROBE_STAMPS
    const long long u2 = wall_clock64();
    if (tid == 0 && (jb.r0 == 0 || jb.r0 == 64))
        printf("update job type %d rows %d: relation products %lld, type term %lld (x10 ns), started %lld\n", jb.idx, jb.r0, u1 - u0, u2 - u1, u0);
#endif
    const T* E = (const T*)ty.E;
    const T* D = (const T*)ty.D;
    T* Gw = (T*)ty.G;
    const T eps = (T)2.220446049250313e-16;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
#pragma unroll
    for (int i = 0; i < SmTile<T>::WR; ++i)
#pragma unroll
        for (int j = 0; j < SmTile<T>::WC; ++j)
#pragma unroll
            for (int q = 0; q < Mfma<T>::NREG; ++q) {
                const int m = wm0 + i * Mfma<T>::MT + Mfma<T>::d_row(lane, q), a = wn0 + j * Mfma<T>::NT + Mfma<T>::d_col(lane);
                if (m >= jb.nr || a >= c) continue;
                const int64_t off = (int64_t)(jb.r0 + m) * c + a;
                T ev = e.acc[i][j][q], dv = d.acc[i][j][q];
                if (ty.has_theta) {
                    ev += E[off];
                    dv += D[off];
                }
                const T den = (dv > eps || dv != dv) ? dv : eps;     // np.maximum(D, eps) (NaN propagates)
                Gw[off] = Gw[off] * (T)sqrt(ev / den);
            }
}

}  // namespace skf
