// skf_api.hip -- C ABI (include/skfusion_hip.h) and the host-side plan / launch schedule of the
// DFMF / DFMC / fold-in iteration.  Kernels: skf_kernels.h.
//
// One iteration (reference _dfmf.py:228-296, 2-GEMM form of SURVEY.md 7.0; everything reads the
// OLD factors, G is replaced at the very end):
//   Gram_i = G_i^T G_i                    split-K MFMA GEMM + fixed-order reduce   (:228-231)
//   K_i    = pinv(Gram_i)                 Jacobi eigen kernel, one workgroup/type  (:232)
//   P_r = R_r G_j ; Q_r = R_r^T G_i       the two big contractions (only reads of R)
//   S_r = K_i (G_i^T P_r) K_j             (:236-239)
//   [DFMC] R_r[mask] = (G_i S_r G_j^T)[mask], P_r recomputed               (_dfmc.py:319-325)
//   E_i += (P_r S_r^T)+ + G_i B-  ; D_i += (P_r S_r^T)- + G_i B+ ,  B = S Gram_j S^T  (:254-281)
//   E_j += (Q_r S_r)+   + G_j D-  ; D_j += (Q_r S_r)-   + G_j D+ ,  D = S^T Gram_i S
//   D_i += Theta+ G_i ; E_i += Theta- G_i                                   (:284-292)
//   G_i <- G_i * sqrt(E_i / max(D_i, eps))                                  (:294-296)
#include "skf_kernels.h"
#include "skf_known.h"
#include "skf_small.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/skfusion_hip.h"

struct skf_comm;

namespace skf {

static thread_local std::string g_err;

struct Error {
    int code;
    std::string msg;
};

#define SKF_FAIL(code_, ...)                                  \
    do {                                                      \
        char buf_[512];                                       \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);             \
        throw Error{(code_), std::string(buf_)};              \
    } while (0)

#define SKF_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) SKF_FAIL(SKF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// the ONE place the library reads its environment (test / A-B switches: Switches::read and plan creation; SKF_RCCL_PATH)
static const char* env_str(const char* name) { return getenv(name); }
static int env_int(const char* name, int unset) {
    const char* v = env_str(name);
    return v ? atoi(v) : unset;
}

static thread_local int64_t g_launches = 0;      // kernel launches this thread has issued through the library (skf_launch_count)

static inline void check_launch(const char* what) {
    ++g_launches;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) SKF_FAIL(SKF_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
}

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return SKF_OK;
    } catch (const Error& e) {
        g_err = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        g_err = e.what();
        return SKF_E_INVALID;
    }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// SKF_OPT_OWNED_ROWS: rows per rank of a type of n objects over `world` ranks (include/skfusion_hip.h, skf_owned_rows):
// boundaries at multiples of the 256-row contraction tile for large types, of 64 for SKF_BF16 (the K padding of Q = R^T G_i)
static inline int64_t owned_chunk(int dtype, int64_t n, int world) {
    const int64_t per = (n + world - 1) / world;
    const int64_t a = per >= 1024 ? 256 : (dtype == 2 /* SKF_BF16 */ ? 64 : 1);
    return (per + a - 1) / a * a;
}
static inline int elem_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;           // grid-stride the rest (256 CUs x 8 blocks)
    return (int)b;
}

// ------------------------------------------------------------------------------------------
// GEMM dispatch
// ------------------------------------------------------------------------------------------
struct TileCfg {
    int bm, bn, bk;
};

template <typename T> struct Tiles;
template <> struct Tiles<float> {
    static constexpr int BK = 32;
    static TileCfg big() { return {128, 128, BK}; }
    static TileCfg small() { return {64, 64, BK}; }
};
// f64: a 128 x 128 tile needs 128 accumulator registers + staging = 376 registers (1 wave per
// SIMD); 64 x 128 (2 x 4 MFMA tiles per wave) fits several waves per SIMD
template <> struct Tiles<double> {
    static constexpr int BK = 16;
    static TileCfg big() { return {64, 128, BK}; }
    static TileCfg small() { return {32, 32, BK}; }
    // c x c x c products of the backbone algebra (c <= 512): latency-bound -- many small tiles over
    // the chip and few, deep K steps instead of 8 workgroups walking 16 shallow ones
    static TileCfg deep() { return {32, 32, 64}; }
};
template <typename T> struct BigWave;
template <> struct BigWave<float> { static constexpr int WR = 2, WC = 2; };     // 128 x 128
template <> struct BigWave<double> { static constexpr int WR = 2, WC = 4; };    //  64 x 128

static TileCfg pick_tile(bool is_f64, int engine, int M, int N, int K = 0, bool all_f64 = false) {
    if (engine == SKF_ENGINE_VALU) return {64, 64, 16};
    if (all_f64 && K >= 64 && K <= 1024 && (int64_t)M * N <= 512 * 512) return Tiles<double>::deep();
    const bool big = (M > 64 && N > 64);
    if (is_f64) return big ? Tiles<double>::big() : Tiles<double>::small();
    return big ? Tiles<float>::big() : Tiles<float>::small();
}

// tiles of a symmetric M x M product that touch the lower triangle (GemmArgs::sym)
static int sym_tiles(const TileCfg& t, int M) {
    int n = 0;
    for (int by = 0; by * t.bm < M; ++by) {
        const int cnt = (by * t.bm + t.bm - 1) / t.bn + 1, all = cdiv(M, t.bn);
        n += cnt < all ? cnt : all;
    }
    return n;
}

// number of K slices: fill the chip (>= ~512 workgroups) when the output has few tiles
static int pick_splits(const TileCfg& t, int M, int N, int K, bool sym = false) {
    const int64_t tiles = sym ? sym_tiles(t, M) : (int64_t)cdiv(M, t.bm) * cdiv(N, t.bn);
    if (tiles >= 256) return 1;
    const int ktiles = cdiv(K, t.bk);
    int s = (int)(512 / tiles);
    if (ktiles < 32) return 1;                                   // short contractions (c x c x c) are not split
    const int max_by_k = ktiles / 8 > 0 ? ktiles / 8 : 1;       // at least 8 K tiles per slice
    if (s > max_by_k) s = max_by_k;
    if (s > 256) s = 256;
    return s < 1 ? 1 : s;
}

// K slices for the relation contractions of the f32 / f64 engines (tiles >= 256: pick_splits leaves them unsplit).  Their
// launches are matrix-core bound and their workgroup counts sit just above a multiple of what the chip holds: config 3's
// 50 000-row relations are 782 tiles of 128 x 128 on 768 slots (3 workgroups per CU), the 100 000-row one 1564 -- 14 / 28
// workgroups run a round of their own (a workgroup alone on its CU runs about three times as fast as one of three, so the
// tail costs a third of a round: 75 % efficiency at one round, 86 % at two).  More slices dilute the tail:
//   time(s) = (full rounds + ceil(tail workgroups / 256) / 3) x (K tiles per slice + 4) x step + s x (partial write + read)
// with the step of a workgroup on a full CU (f32 128 x 128 x 32: 6.5 us, f64 64 x 128 x 16: 3.7 us).  Round 5, config 3:
// f32 engine 9.9 -> 11.4 it/s (P12: 5 slices).
static int pick_splits_relation(const TileCfg& t, int M, int N, int K, bool is_f64) {
    const int64_t tiles = (int64_t)cdiv(M, t.bm) * cdiv(N, t.bn);
    const int ktiles = cdiv(K, t.bk);
    const double occ = 3.0, slots = 256.0 * occ, step_us = is_f64 ? 3.7 : 6.5;
    const double slice_us = (double)M * (double)N * (is_f64 ? 8.0 : 4.0) * 2.0 / 4.0e6;      // partials written and re-read at ~4 TB/s
    int best = 1;
    double best_t = 1e300;
    for (int sl = 1; sl <= 16; ++sl) {
        if (sl > 1 && ktiles / sl < 64) break;
        const double wgs = (double)tiles * sl;
        const double full = (double)(int64_t)(wgs / slots), rem = wgs - full * slots;
        const double rounds = full + (rem > 0.0 ? (double)(int64_t)((rem + 255.0) / 256.0) / occ : 0.0);
        const double tt = rounds * (cdiv(ktiles, sl) + 4) * step_us + (sl > 1 ? sl * slice_us : 0.0);
        if (tt < best_t * 0.97) {
            best_t = tt;
            best = sl;
        }
    }
    return best;
}

// operand / result types of one contraction (SKF_F64 or SKF_F32 each).  Supported:
//   (f64,f64,f64)  f64 engine, and the c x c algebra of every engine
//   (f32,f32,f32)  relation contractions and Theta products of the f32 engine
//   (f64,f32,f32)  Gram = G^T G and W = G^T P of the f32 engine: f32 operands, f64 arithmetic
//   (f32,f32,f64)  n x c x c products of the f32 engine with an f64 backbone / B, D matrix
struct GemmTypes {
    int c, a, b;
};

// the staging mode the kernel's stage_mode() would pick for an operand, evaluated on the host (every K slice starts at a
// multiple of k_chunk)
template <typename TS>
static int host_stage_mode(const void* src, int64_t s_row, int64_t s_k, int row_end, int K, int k_chunk) {
    constexpr int V = 16 / (int)sizeof(TS);
    const bool aligned = (((uintptr_t)src) & 15) == 0;
    if (s_k == 1 && aligned && s_row % V == 0 && K % V == 0 && k_chunk % V == 0) return STAGE_VEC_K;
    if (s_row == 1 && aligned && s_k % V == 0 && row_end % V == 0) return STAGE_VEC_R;
    return STAGE_SCALAR;
}

// The big tile stages its operands with COMPILE-TIME modes (with run-time modes the unrolled staging code of both forms
// ran the f32 kernels out of registers: 15 - 57 spilled per lane).  The layouts the engines produce have instantiations:
//   K|R  A along K, B along its rows   P = R G_j, Theta G, H = G_i S, the n x c x c side products
//   R|R  both along their rows         Q = R^T G_i, Gram = G^T G, W = G_i^T P
//   K|K  both along K                  the reconstruction H G_j^T of the f32 / f64 completion and residual
// (every K slice starts at a multiple of the K tile, so the slicing never changes the verdict).  Returns the mode pair, or
// -1: no instantiation -- such a product runs on the small tile, whose run-time modes cost no registers that matter.
constexpr int FM_KR = STAGE_VEC_K | (STAGE_VEC_R << 2), FM_RR = STAGE_VEC_R | (STAGE_VEC_R << 2), FM_KK = STAGE_VEC_K | (STAGE_VEC_K << 2);
static int big_tile_modes(GemmTypes ty, const GemmArgs& g, bool relation) {
    const int bk = (ty.c == SKF_F64) ? Tiles<double>::BK : Tiles<float>::BK;
    const int ma = ty.a == SKF_F64 ? host_stage_mode<double>(g.A, g.sa_m, g.sa_k, g.M, g.K, bk)
                                   : host_stage_mode<float>(g.A, g.sa_m, g.sa_k, g.M, g.K, bk);
    const int mb = ty.b == SKF_F64 ? host_stage_mode<double>(g.B, g.sb_n, g.sb_k, g.N, g.K, bk)
                                   : host_stage_mode<float>(g.B, g.sb_n, g.sb_k, g.N, g.K, bk);
    const int fm = ma | (mb << 2);
    const bool same = ty.a == ty.c && ty.b == ty.c;
    if (same) return (fm == FM_KR || fm == FM_RR || (fm == FM_KK && !relation)) ? fm : -1;
    if (ty.c == SKF_F64) return fm == FM_RR ? fm : -1;           // (f64, f32, f32)
    return fm == FM_KR ? fm : -1;                                // (f32, f32, f64)
}
// the tile a product runs on (run_gemm, and callers that size per-workgroup outputs: skf_relation_sqerr)
static TileCfg gemm_tile(GemmTypes ty, int engine, const GemmArgs& g, bool deep_ok, bool relation) {
    const bool is_f64 = (ty.c == SKF_F64);
    const TileCfg t = pick_tile(is_f64, engine, g.M, g.N, g.K, deep_ok);
    if (engine != SKF_ENGINE_MFMA) return t;
    const TileCfg big = is_f64 ? Tiles<double>::big() : Tiles<float>::big();
    if (t.bm == big.bm && t.bn == big.bn && t.bk == big.bk && big_tile_modes(ty, g, relation) < 0)
        return is_f64 ? Tiles<double>::small() : Tiles<float>::small();
    return t;
}

template <typename T, typename TA, typename TB>
static void launch_gemm_t(int engine, const TileCfg& t, GemmArgs g, int splits, bool relation, hipStream_t st) {
    dim3 grid(cdiv(g.N, t.bn), cdiv(g.M, t.bm), splits);
    if (g.sym) grid = dim3((unsigned)sym_tiles(t, g.M), 1, splits);      // (GemmArgs::sym: the tiles on / below the diagonal)
    dim3 block(GEMM_THREADS);
    constexpr int WRB = BigWave<T>::WR, WCB = BigWave<T>::WC;
    const bool big = (t.bm == Tiles<T>::big().bm && t.bn == Tiles<T>::big().bn);
    if (engine == SKF_ENGINE_VALU) {
        hipLaunchKernelGGL((gemm_valu_kernel<T, TA, TB>), grid, block, 0, st, g);
    } else if (t.bk == 64 && t.bm == 32) {         // Tiles<double>::deep()
        if constexpr (std::is_same<T, double>::value && std::is_same<TA, double>::value &&
                      std::is_same<TB, double>::value)
            hipLaunchKernelGGL((gemm_mfma_kernel<double, double, double, 1, 1, 64, 0>), grid, block, 0, st, g);
        else
            SKF_FAIL(SKF_E_INVALID, "deep tile is f64 only");
    } else if (big) {
        constexpr int BKT = Tiles<T>::BK;
        constexpr bool same = std::is_same<TA, T>::value && std::is_same<TB, T>::value;
        const int fm = big_tile_modes(GemmTypes{std::is_same<T, double>::value ? SKF_F64 : SKF_F32, std::is_same<TA, double>::value ? SKF_F64 : SKF_F32,
                                                std::is_same<TB, double>::value ? SKF_F64 : SKF_F32}, g, relation);
        if (fm < 0) SKF_FAIL(SKF_E_INVALID, "big tile without a staging layout (gemm_tile picks the small tile for these)");
#define SKF_BIG(TAG_, FM_) hipLaunchKernelGGL((gemm_mfma_kernel<T, TA, TB, WRB, WCB, BKT, TAG_, FM_>), grid, block, 0, st, g)
        if constexpr (same) {
            if (relation) {
                if (fm == FM_KR) SKF_BIG(1, FM_KR);
                else SKF_BIG(1, FM_RR);
            } else {
                if (fm == FM_KR) SKF_BIG(0, FM_KR);
                else if (fm == FM_RR) SKF_BIG(0, FM_RR);
                else SKF_BIG(0, FM_KK);
            }
        } else if constexpr (std::is_same<T, double>::value) {      // (f64, f32, f32): Gram = G^T G and W = G_i^T P of the f32 / bf16 engines
            SKF_BIG(0, FM_RR);
        } else {                                                     // (f32, f32, f64): n x c x c products with an f64 backbone
            SKF_BIG(0, FM_KR);
        }
#undef SKF_BIG
    } else {
        hipLaunchKernelGGL((gemm_mfma_kernel<T, TA, TB, 1, 1, Tiles<T>::BK, 0>), grid, block, 0, st, g);
    }
    check_launch("gemm");
    if (splits > 1) {
        if (splits >= 8)
            hipLaunchKernelGGL((splitk_reduce_z16_kernel<T>), dim3(elem_grid((int64_t)g.M * g.N * 16)), dim3(256), 0, st,
                               g, splits);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(elem_grid((int64_t)g.M * g.N)), dim3(256), 0, st,
                               g, splits);
        check_launch("splitk_reduce");
    }
}

// `part`/`part_bytes`: scratch for split-K partials; splits is clamped to fit.
static void run_gemm(GemmTypes ty, int engine, GemmArgs g, int want_splits, void* part, size_t part_bytes,
                     hipStream_t st, bool relation = false) {
    if (g.M <= 0 || g.N <= 0) return;
    const bool is_f64 = (ty.c == SKF_F64);
    const bool all_f64 = (ty.c == SKF_F64 && ty.a == SKF_F64 && ty.b == SKF_F64);
    const TileCfg t = gemm_tile(ty, engine, g, all_f64 && want_splits <= 1, relation);
    const bool sym_ok = g.sym && g.M == g.N && engine == SKF_ENGINE_MFMA && t.bn > t.bm && g.epi == EPI_STORE;
    int splits = want_splits > 0 ? want_splits : pick_splits(t, g.M, g.N, g.K, sym_ok);
    if (want_splits <= 0 && relation && engine == SKF_ENGINE_MFMA && g.epi == EPI_STORE && t.bm >= 64 &&
        (int64_t)cdiv(g.M, t.bm) * cdiv(g.N, t.bn) >= 256) {
        splits = pick_splits_relation(t, g.M, g.N, g.K, is_f64);
    }
    if (g.epi == EPI_SQDIFF) splits = 1;
    const size_t per = (size_t)g.M * g.N;
    const size_t part_elems = part_bytes / (is_f64 ? 8 : 4);
    if (splits > 1 && (part == nullptr || per * splits > part_elems)) {
        splits = part ? (int)(part_elems / per) : 1;
        if (splits < 1) splits = 1;
    }
    int ktiles = cdiv(g.K > 0 ? g.K : 1, t.bk);
    if (splits > ktiles) splits = ktiles;
    g.k_chunk = cdiv(ktiles, splits) * t.bk;
    splits = cdiv(g.K > 0 ? g.K : 1, g.k_chunk);
    g.part = part;
    // a symmetric product (the caller says so: Gram = G^T G) computes the tiles on and below the diagonal only; the reduce of
    // the K slices mirrors the rest (GemmArgs::sym).  Unsplit launches write C themselves and compute every tile.
    g.sym = (sym_ok && splits > 1) ? (t.bm | (t.bn << 16)) : 0;
    if (ty.c == SKF_F64 && ty.a == SKF_F64 && ty.b == SKF_F64)
        launch_gemm_t<double, double, double>(engine, t, g, splits, relation, st);
    else if (ty.c == SKF_F32 && ty.a == SKF_F32 && ty.b == SKF_F32)
        launch_gemm_t<float, float, float>(engine, t, g, splits, relation, st);
    else if (ty.c == SKF_F64 && ty.a == SKF_F32 && ty.b == SKF_F32)
        launch_gemm_t<double, float, float>(engine, t, g, splits, relation, st);
    else if (ty.c == SKF_F32 && ty.a == SKF_F32 && ty.b == SKF_F64)
        launch_gemm_t<float, float, double>(engine, t, g, splits, relation, st);
    else
        SKF_FAIL(SKF_E_INVALID, "unsupported operand type combination (c=%d a=%d b=%d)", ty.c, ty.a, ty.b);
}

// Two independent all-f64 products of the c x c chains in ONE launch (gemm_mfma_pair_kernel) when both take the deep
// unsplit tile of the matrix-core engine -- what run_gemm picks for them one by one; anything else: two launches.
static void run_gemm_pair_f64(int engine, GemmArgs a, GemmArgs b, hipStream_t st, bool allow) {
    const GemmTypes ty{SKF_F64, SKF_F64, SKF_F64};
    auto deep = [&](const GemmArgs& g) {
        if (g.M <= 0 || g.N <= 0 || g.sym || g.epi == EPI_SQDIFF) return false;
        const TileCfg t = gemm_tile(ty, engine, g, true, false);
        return engine == SKF_ENGINE_MFMA && t.bk == 64 && t.bm == 32 && pick_splits(t, g.M, g.N, g.K, false) == 1;
    };
    if (!allow || !deep(a) || !deep(b)) {
        run_gemm(ty, engine, a, 0, nullptr, 0, st);
        run_gemm(ty, engine, b, 0, nullptr, 0, st);
        return;
    }
    const TileCfg t = gemm_tile(ty, engine, a, true, false);
    for (GemmArgs* g : {&a, &b}) {                  // (as run_gemm sets an unsplit launch up)
        g->k_chunk = cdiv(g->K > 0 ? g->K : 1, t.bk) * t.bk;
        g->part = nullptr;
        g->sym = 0;
    }
    const int gx = std::max(cdiv(a.N, t.bn), cdiv(b.N, t.bn)), gy = std::max(cdiv(a.M, t.bm), cdiv(b.M, t.bm));
    hipLaunchKernelGGL((gemm_mfma_pair_kernel<double, double, double, 1, 1, 64, 0>), dim3(gx, gy, 2), dim3(GEMM_THREADS), 0, st, a, b);
    check_launch("gemm_pair");
}

// ---- bf16 relation contraction --------------------------------------------------------------
static inline int64_t pad64(int64_t v) { return (v + 63) / 64 * 64; }

// K slices for the bf16 contraction.  Time model of a launch (microseconds): the workgroups run in
// ceil(units * s / slots) rounds of (K tiles per slice + 3 tiles of prologue / epilogue) x 1.5 us, and s > 1 slices
// write and re-read s partial copies of the output at ~3 TB/s.  On config 3 it picks what the A/B runs of rounds 1 / 2
// picked (5 / 3 / 3 / 3 slices for P12 / Q12 / P23 / Q23); a product with a handful of output tiles (config 5,
// genre x movie: ONE 256 x 256 tile over K = 40000, 0.96 ms in one workgroup before) is cut into up to 64 slices.
static int pick_splits_bf16(int64_t units, int ktiles, int bm, int64_t out_elems) {
    const double cus = 256.0;                                       // resident workgroups: one (256 rows) or two per CU
    const double slots = cus * (bm >= 256 ? 1.0 : 2.0);
    const double per_slice_us = (double)out_elems * 8.0 / 3.0e6;
    int best = 1;
    double best_t = 1e300;
    for (int s = 1; s <= 64; ++s) {
        if (s > 1 && ktiles / s < 4) break;
        const double rounds = (double)(int64_t)((double)units * s / slots + 0.999999);
        const double t = (rounds < 1.0 ? 1.0 : rounds) * ((ktiles + s - 1) / s + 3) * 1.5 + (s > 1 ? s * per_slice_us : 0.0);
        // more slices only for a clear gain (HBM-bound launches have none).  Round 3, profiles/r03_split_margin.txt: with the
        // margin at 1.0 the model cuts P13 (196 of 256 CUs busy, one slice) into five slices -- inside the iteration that
        // launch then takes 1.23 ms instead of 1.19 and the fit loses 0.7 %: the 60 idle CUs are not idle, the side
        // products of the second stream run there.
        if (t < best_t * 0.97) {
            best_t = t;
            best = s;
        }
    }
    return best;
}

// rows per workgroup of the bf16 contraction: the 256-row LDS-DMA kernel for large problems and for
// every transposed-A product, the 128-row register-staged kernel for small P-type products
static int bf16_block_rows(int M, bool at) { return (at || M >= 4096) ? 256 : 128; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and DEVICE (the attribute belongs to the kernel's
// code object on one device: a process that drives several GPUs sets it on each), safe against the concurrent host
// threads of run_fits_concurrent (one plan per thread)
constexpr int SKF_MAX_DEVICES = 64;
constexpr int SKF_MAX_BATCH = 64;        // plans of one skf_iterate_batch call
struct DeviceOnce {
    std::once_flag flag[SKF_MAX_DEVICES];
};
template <class K>
static void allow_dynamic_lds(DeviceOnce& once, K kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SKF_MAX_DEVICES) dev = 0;
    hipError_t err = hipSuccess;
    std::call_once(once.flag[dev], [&] { err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
    if (err != hipSuccess) SKF_FAIL(SKF_E_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(err));
}

// C[M x N] (f32) = op(A) * Bt^T, bf16 operands:  at == false: A is [M][lda] (K contiguous);
// at == true: A is [Kp][lda] row-major with the OUTPUT rows along its columns (lda >= M, rows zero-padded to Kp)
static void run_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* Bt, int64_t ldb, float* C, int64_t ldc,
                          int M, int N, int Kp, int want_splits, void* part, size_t part_bytes, bool relation,
                          hipStream_t st, bool at = false, bool abits = false) {
    if (M <= 0 || N <= 0) return;
    // abits: A is the bitmap of a binary relation, lda its row pitch in bytes (8 entries per byte)
    const int64_t a_cols = abits ? lda * 8 : lda;
    if (Kp % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || (!at && a_cols < Kp) || (at && a_cols < M) || ldb < Kp)
        SKF_FAIL(SKF_E_INVALID, "bf16 contraction: inner dimension must be padded to 64 (Kp=%d lda=%lld ldb=%lld)", Kp,
                 (long long)lda, (long long)ldb);
    if ((((uintptr_t)A) | ((uintptr_t)Bt)) & 15) SKF_FAIL(SKF_E_INVALID, "bf16 operands must be 16-byte aligned");
    const int bn = (N <= 128) ? 128 : 256;
    const int bm = bf16_block_rows(M, at || abits);
    const int ktiles = Kp / 64;
    const int64_t units = (int64_t)cdiv(M, bm) * cdiv(N, bn);
    int splits = want_splits > 0 ? want_splits : pick_splits_bf16(units, ktiles, bm, (int64_t)M * N);
    const size_t per = (size_t)M * N * sizeof(float);
    if (splits > 1 && (!part || per * splits > part_bytes)) splits = part ? (int)(part_bytes / per) : 1;
    if (splits < 1) splits = 1;
    if (splits > ktiles) splits = ktiles > 0 ? ktiles : 1;
    Bf16GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = A; g.Bt = Bt; g.C = C; g.part = (float*)part;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.Kp = Kp;
    g.a_kstep = 64; g.b_kstep = 64;
    g.k_chunk = cdiv(ktiles > 0 ? ktiles : 1, splits) * 64;
    splits = cdiv(Kp > 0 ? Kp : 1, g.k_chunk);
    dim3 grid(cdiv(N, bn), cdiv(M, bm), splits);
    if (bm == 256) {
        // 256 x BN tile, LDS rings in dynamic shared memory (> 64 KiB needs the attribute)
        // (SKF_V2_POLICY: what the K loop does -- the product's V2Full unless a bound-finding build of tools/probe names
        // a policy of its own on the command line, the way SKF_A_AUX names the cache policy)
#ifndef SKF_V2_POLICY
#define SKF_V2_POLICY V2Full
#endif
#define SKF_V2_LAUNCH(BN_, TAG_, AT_)                                                                             \
    do {                                                                                                          \
        const int smem_ = (3 * 256 + ((BN_ == 256) ? 2 : 3) * BN_) * 8 * 16;                                      \
        static DeviceOnce once_;                                                                              \
        allow_dynamic_lds(once_, gemm_bf16_v2_kernel<BN_, TAG_, AT_, EPI_T_STORE, false, SKF_V2_POLICY>, smem_);  \
        hipLaunchKernelGGL((gemm_bf16_v2_kernel<BN_, TAG_, AT_, EPI_T_STORE, false, SKF_V2_POLICY>), grid, dim3(512), smem_, st, g); \
    } while (0)
#define SKF_V2_LAUNCH_BITS(BN_, AT_)                                                                              \
    do {                                                                                                          \
        const int smem_ = (3 * 256 + ((BN_ == 256) ? 2 : 3) * BN_) * 8 * 16;                                      \
        static DeviceOnce once_;                                                                              \
        allow_dynamic_lds(once_, gemm_bf16_v2_kernel<BN_, 1, AT_, EPI_T_STORE, true>, smem_);                     \
        hipLaunchKernelGGL((gemm_bf16_v2_kernel<BN_, 1, AT_, EPI_T_STORE, true>), grid, dim3(512), smem_, st, g); \
    } while (0)
        if (abits) {
            if (at && bn == 128) SKF_V2_LAUNCH_BITS(128, true);
            else if (at) SKF_V2_LAUNCH_BITS(256, true);
            else if (bn == 128) SKF_V2_LAUNCH_BITS(128, false);
            else SKF_V2_LAUNCH_BITS(256, false);
        } else if (at) {
            if (bn == 128) SKF_V2_LAUNCH(128, 1, true);
            else SKF_V2_LAUNCH(256, 1, true);
        } else if (bn == 128 && relation) SKF_V2_LAUNCH(128, 1, false);
        else if (bn == 128) SKF_V2_LAUNCH(128, 0, false);
        else if (relation) SKF_V2_LAUNCH(256, 1, false);
        else SKF_V2_LAUNCH(256, 0, false);
#undef SKF_V2_LAUNCH
#undef SKF_V2_LAUNCH_BITS
    } else {
        dim3 block(256);
        if (bn == 128) hipLaunchKernelGGL((gemm_bf16_kernel<128, 0>), grid, block, (128 + 128) * 128, st, g);
        else hipLaunchKernelGGL((gemm_bf16_kernel<256, 0>), grid, block, (128 + 256) * 128, st, g);
    }
    check_launch("gemm_bf16");
    if (splits > 1) {
        hipLaunchKernelGGL(bf16_splitk_reduce_kernel, dim3(elem_grid(((int64_t)M * N + 3) / 4)), dim3(256), 0, st, C, ldc,
                           (const float*)part, M, N, splits);
        check_launch("bf16_splitk_reduce");
    }
}

static size_t bf16_part_bytes(int M, int N, int Kp, bool at) {      // (a bitmap operand always takes the 256-row kernel)
    const int bn = (N <= 128) ? 128 : 256;
    const int bm = bf16_block_rows(M, at);
    const int s = pick_splits_bf16((int64_t)cdiv(M, bm) * cdiv(N, bn), Kp / 64, bm, (int64_t)M * N);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

template <typename TS>
static void launch_to_bf16(uint16_t* dst, int64_t ldd, const TS* src, int64_t lds, int64_t rows, int64_t cols,
                           bool transpose, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return;
    if (transpose) {
        dim3 grid((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32));
        hipLaunchKernelGGL((transpose_to_bf16_kernel<TS>), grid, dim3(256), 0, st, dst, ldd, src, lds, rows, cols);
    } else {
        hipLaunchKernelGGL((to_bf16_kernel<TS>), dim3(elem_grid(rows * cols)), dim3(256), 0, st, dst, ldd, src, lds,
                           rows, cols);
    }
    check_launch("to_bf16");
}

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
// Test / A-B switches from the environment.  A plan reads them ONCE, when its workspace is bound (skf::Switches::read);
// no launch path reads the environment.  Defaults are the measured best.
struct Switches {
    bool pinv_jacobi = false;      // SKF_PINV_JACOBI=1     every pseudo-inverse through the Jacobi eigen-solver
    bool chol_unblocked = false;   // SKF_CHOL_UNBLOCKED=1  plain (unblocked) Cholesky inverse
    bool chol_no_small = false;    // SKF_CHOL_NO_SMALL=1   no one-wave kernel for orders <= 64
    bool no_small_chain = false;   // SKF_NO_SMALL_CHAIN=1  general c x c launches instead of the one-workgroup chains
    bool debug_pinv = false;       // SKF_DEBUG_PINV=1      per iteration: verdict of the fast path (stderr; synchronises)
    bool graph = false;            // SKF_GRAPH=1           hipGraph replay of the iteration
    bool no_overlap = false;       // SKF_NO_OVERLAP=1      no second stream
    bool no_pipeline = false;      // SKF_NO_PIPELINE=1     staged schedule instead of the relation pipeline
    int side_tile = 0;             // SKF_SIDE_TILE=64|128  tile shape of the fused side update
    bool no_small_fused = false;   // SKF_NO_SMALL_FUSED=1  small graphs on the general staged schedule (~33 launches per iteration)
    bool no_sweep = false;         // SKF_PINV_SWEEP=0      orders 65 .. 256: blocked Cholesky inverse + unpack instead of the blocked sweep (A/B)
    bool small_sweep1 = false;     // SKF_SMALL_SWEEP4=0    small graphs: one pivot per barrier in the register sweep (same bits; A/B, tests)
    int gram_sym = 1;              // SKF_GRAM_SYM=0        split-K Gram products compute every tile (default: the tiles on / below the diagonal, mirrored by the reduce)
    int sweep_step_min = 65;       // SKF_SWEEP_STEP_MIN=n  orders >= n take the blocked sweep one launch per block step with the update of a step
                                   //                       spread over row slabs (sweep_step_kernel); below: one workgroup per matrix, one launch.
                                   //                       0 = never (A/B; same bits either way)
    int sweep_rows = 32;           // SKF_SWEEP_ROWS=32|64.. rows of a slab (a multiple of 32)
    bool no_pairs = false;         // SKF_CHAIN_PAIRS=0     the independent c x c products of a relation's chain as launches of their own (A/B; same bits)
    bool no_sweep_big = false;     // SKF_SWEEP_BIG=0       orders above 256 on the blocked Cholesky inverse + unpack, as before round 5 (A/B)
    int dfmc_sparse = -1;          // SKF_DFMC_SPARSE=0|1   masked relations never / whenever a bound is given as lists of their known entries
    int known_parts = 0;           // SKF_KNOWN_PARTS=1|2|4|8  parts of the known-entry lists (0: by the size of the gathered matrix)
    bool known_parts_forced = false;   // ... given at all: short lists are cut into parts too (tests)
    bool comm_stream = true;       // SKF_COMM_STREAM=0     plans with owned rows: the exchanges on the main stream (A/B; tests run both)
    bool early_update = true;      // SKF_EARLY_UPDATE=0    pipeline: every type is updated at the end of the iteration (default: a type whose last relation is through
                                   //                       and that nothing reads any more is updated on the second stream, underneath the remaining contractions)
    static Switches read() {
        auto on = [](const char* name) { return env_int(name, 0) != 0; };
        auto off = [](const char* name) { return env_int(name, 1) == 0; };         // "=0" switches a default off
        Switches w;
        w.pinv_jacobi = on("SKF_PINV_JACOBI");
        w.chol_unblocked = on("SKF_CHOL_UNBLOCKED");
        w.chol_no_small = on("SKF_CHOL_NO_SMALL");
        w.no_small_chain = on("SKF_NO_SMALL_CHAIN");
        w.debug_pinv = on("SKF_DEBUG_PINV");
        w.graph = on("SKF_GRAPH");
        w.no_overlap = on("SKF_NO_OVERLAP");
        w.no_pipeline = on("SKF_NO_PIPELINE");
        w.no_small_fused = on("SKF_NO_SMALL_FUSED");
        w.no_sweep = off("SKF_PINV_SWEEP");
        w.small_sweep1 = off("SKF_SMALL_SWEEP4");
        w.gram_sym = off("SKF_GRAM_SYM") ? 0 : 1;
        w.early_update = !off("SKF_EARLY_UPDATE");
        w.sweep_step_min = env_int("SKF_SWEEP_STEP_MIN", w.sweep_step_min);
        { const int sr = env_int("SKF_SWEEP_ROWS", 0); if (sr >= 32) w.sweep_rows = (sr + 31) / 32 * 32; }
        w.no_sweep_big = off("SKF_SWEEP_BIG");
        w.no_pairs = off("SKF_CHAIN_PAIRS");
        w.side_tile = env_int("SKF_SIDE_TILE", 0);
        w.dfmc_sparse = env_int("SKF_DFMC_SPARSE", -1);
        { const int kp = env_int("SKF_KNOWN_PARTS", 0); w.known_parts = (kp == 1 || kp == 2 || kp == 4 || kp == 8) ? kp : 0; }
        w.known_parts_forced = env_str("SKF_KNOWN_PARTS") != nullptr;
        w.comm_stream = !off("SKF_COMM_STREAM");
        return w;
    }
};

struct Slot {            // a workspace sub-allocation
    size_t off = 0, bytes = 0;
    void* ptr = nullptr;
};

struct TypeState {
    int64_t n = 0;
    int c = 0;
    int n_pad = 0;                 // eigen order (even)
    Slot G, E, D, Gram, K;
    Slot Bp32, Bn32;               // f32 engines: roundings of Bp_tot / Bn_tot for the fused side update
    Slot Bp_tot, Bn_tot;           // sum over the relations of the B / D matrices' + and - parts (c x c f64)
    Slot Ec, Dc;                   // SKF_TRANSFORM target only
    Slot Galt;                     // SKF_TRANSFORM target: the second factor buffer of the one-launch iteration (foldin_step_kernel)
    int64_t t0 = 0, tn = 0;        // rows whose type-level terms G (sum B) this plan adds (row-block sharding)
    // SKF_OPT_OWNED_ROWS: [t0, t0 + tn) are the rows this plan OWNS; `chunk` rows per rank in the padded layout of the
    // exchanges (G, E, D, the bf16 rows and every Q over this type hold part_count * chunk rows)
    int64_t chunk = 0, n_alloc = 0;
    bool gather_master = true;     // the all-gather of the updated rows carries the master copy (false: SKF_BF16 operand rows only)
    Slot GTb;                      // SKF_BF16: bf16 transpose of G, [c][pad64(n)], zero padded
    int64_t ldgt = 0;
    bool set = false;
    bool keep_prev = false;        // a known-entries relation touches this type: Gp = the factor before the last update
    Slot Gp;
    // SKF_BF16: the bf16 ROWS of G, [n + 1][ldrow] with an all-zero last row -- the gathered matrix of the list passes over
    // the known entries of a masked relation and over the ones of a sparse 0/1 relation (skf_known.h); kept in step with GTb
    bool need_rows = false;
    Slot Grow;
    int64_t ldrow = 0;
};

struct RelState {
    int row = 0, col = 0;
    const void* R_in = nullptr;
    int64_t ld_in = 0;
    const uint8_t* mask = nullptr;
    int64_t ldmask = 0;
    const void* R = nullptr;       // matrix the iteration reads (R_in or the DFMC working copy)
    int64_t ldr = 0;
    Slot Rw, P, Q, W, T1, S, U, H;
    Slot S32;                      // f32 engines: rounding of S for the fused side update
    Slot Hb, Gb;                   // SKF_BF16: bf16 H = G_i S and bf16 G_j (K padded to 64) for the completion / residual tiles
    int64_t ldhb = 0;
    Slot Rb;                       // SKF_BF16: the ONE stored copy of the relation, bf16 [pad64(nr)][pad64(n_j)], zero padded
    int64_t ldrb = 0;
    int64_t kq = 0;                // pad64(nr): inner dimension of Q = R^T G_i (padding rows of Rb are zero)
    bool binary = false;           // SKF_BF16 + SKF_REL_BINARY: the relation is stored as a bitmap (Bb) instead of Rb
    Slot Bb;                       // bitmap [pad64(nr)][ldbb bytes], bit (c & 7) of byte c >> 3; padding zero
    int64_t ldbb = 0;
    // a very sparse binary relation (at most 1 entry in 256 set): the positions of the ones as CSR and CSC beside the bitmap
    bool sparse = false;
    Slot SpRp, SpCi, SpCp, SpRi, SpCnt;     // row pointers / columns, column pointers / rows (ascending), count scratch
    int64_t sp_cap = 0, sp_nnz = 0;
    // ... contracted by srp_bf16_v6_kernel<.., SRP_ONES> over bf16 factor rows when both ranks are 64 / 128 / 256 (up to 1
    // entry in 80 set); otherwise by binary_spmm_kernel over the f32 rows (up to 1 in 256).  Lists in parts pinned to XCDs:
    bool sp_gather = false;
    int sp_pc = 1, sp_pr = 1;               // column parts of the row lists (P), row parts of the column lists (Q)
    int64_t sp_pw = 0, sp_ph = 0;
    Slot SpRpP, SpCpP;                      // segment pointers of the parted lists
    Slot Mb;                       // DFMC: the mask as packed bits, [nr][ldmb bytes], bit (n & 7) of byte n >> 3
    int64_t ldmb = 0;
    bool mask_is_bits = false;     // the caller's mask is already packed (SKF_REL_MASK_BITS)
    Slot Kcnt, Koff, Klist;        // SKF_BF16 masked relation: known entries per 256 x 256 tile (counts, offsets, entries)
    size_t kcap = 0;               // capacity of Klist in entries
    bool use_klist = false;        // the lists fit: the completion writes whole tiles and never reads R back
    bool s_set = false;
    // row-block sharding: this plan holds rows [r0, r0 + nr) of the relation (nr == n_i: all of it)
    int64_t r0 = 0, nr = 0;
    bool absent = false;           // no local rows (W, Q, S are still kept for the exchange)
    bool masked = false;           // DFMC: the relation has a mask (here or, for an absent one, elsewhere)
    bool col_side = true;          // this plan adds the column-side terms E_j, D_j
    // DFMC on the known entries only (skf_relation_desc.known_bound, skf_known.h): no completed copy of the relation
    bool kn = false;
    int64_t kn_cap = 0, kn_nnz = 0;        // bound given at plan creation / entries found at bind time
    int kn_pc = 1, kn_pr = 1;              // column parts of the row lists, row parts of the column lists
    int64_t kn_pw = 0, kn_ph = 0;          // columns per column part (multiple of 64), rows per row part
    int64_t kn_ldf = 0;                    // leading dimension of the gathered vectors (bf16 copies: padded to 8)
    Slot KrPtr, KrIdx, KrVal;              // rows -> known columns (ascending), R there
    Slot KcPtr, KcIdx, KcVal, KcE;         // columns -> known rows (ascending), R there, residuals E of the last iteration
    Slot KCnt;                             // count / fill-position scratch of the bind-time build
    Slot FiB;                              // SKF_BF16: bf16 rows of T = G_j S^T (n_j x ldf; the rows of G_i: TypeState::Grow)
    Slot Tm;                               // T = G_j S^T in the master type (n_j x c_i)
    Slot Apart, Qpart;                     // partial outputs of the parts, [parts][n][c_i] (only with more than one part)
    Slot A;                                // E T, then the row-side product P S^T = G_i (S Gram_j S^T) + E T   (n_i x c_i)
    Slot Sp, Xi, Xj, Bf, U2;               // S of the stored residuals; G_i'^T G_i, G_j^T G_j'; S Gram_j S^T; S^T Gram_i  (f64)
    Slot SmBp, SmBn, SmDp, SmDn;           // small graphs: +- parts of S Gram_j S^T and S^T Gram_i S of THIS relation
    Slot SmQ;                              // ... and Q as shares over row ranges of the relation, [sm_qparts][n_j][c_i]
    int sm_qparts = 0;
};

struct ThetaState {
    int type = 0;
    const void* data = nullptr;
    int64_t ld = 0;
    bool has_pos = true, has_neg = true;   // non-empty halves of the +- split (found at bind time)
    Slot Pb, Nb;                           // SKF_BF16: bf16 copies of Theta+ / Theta-, [n][pad64(n)]
    int64_t ldb = 0;
    // sparse form (skf_theta_desc.nnz > 0): CSR in the master type, built at bind time
    bool sparse = false;
    int64_t nnz_cap = 0, nnz = 0;
    Slot Rp, Ci, Vv, Cnt;                  // rowptr (n + 1, int64), column indices (int32), values, per-row counts
};

}  // namespace skf

struct skf_plan {
    int dtype = SKF_F32, variant = SKF_DFMF, target = -1, engine = SKF_ENGINE_MFMA;
    bool f64 = false, bf16 = false;
    size_t esz = 4;            // bytes of a master element (factors, E, D, P, Q, relations)
    int mt = SKF_F32;          // master type code; the c x c algebra is always SKF_F64
    std::vector<skf::TypeState> types;
    std::vector<skf::RelState> rels;
    std::vector<skf::ThetaState> thetas;
    std::vector<skf::Slot*> slots;
    size_t ws_bytes = 0;
    bool bound = false, prepared = false, first_iter = true;
    // shared scratch
    skf::Slot part;            // split-K partials
    size_t part_bytes = 0;
    skf::Slot eigA, eigV, eigVs, eigW, eigN, eigNorig, eigOk, sqpart;
    skf::Slot eigX;                        // scratch of the multi-workgroup deflation (plans with a rank above 256)
    skf::Slot theta_flags, theta_tmp;      // sign flags of the constraints; SKF_BF16: n x c product scratch
    int64_t eig_stride = 0;
    int eig_maxn = 0;
    size_t sq_elems = 0;
    // second stream: Gram + pseudo-inverse run concurrently with the relation contractions
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    skf::Slot part_aux;
    size_t part_aux_bytes = 0;
    skf::Slot sp_part;                     // partial outputs of the parted list passes over sparse 0/1 relations

    skf::Switches sw;                      // read once in skf_plan_bind_workspace
    bool overlap = false;
    bool pipeline = true;                  // relation-pipelined schedule of the DFMF iteration (SKF_NO_PIPELINE=1 at bind: off)
    std::vector<hipEvent_t> ev_rel;        // one event per relation: its contractions are done
    size_t acc_off = 0, acc_bytes = 0;     // contiguous range of all E / D accumulators
    // E, D and G of all types as three regions of identical layout (flat_bytes each, a pad behind every one)
    size_t flat_e_off = 0, flat_d_off = 0, flat_g_off = 0, flat_bytes = 0;
    skf::Slot flat_pad[3];
    skf_comm* comm = nullptr;              // collectives of the distributed iteration (skf_plan_set_comm; not owned)
    // SKF_OPT_OWNED_ROWS: ownership-aligned row blocks (iterate_owned)
    bool owned = false;
    int part_index = 0, part_count = 0;
    size_t xg_off = 0, xg_bytes = 0;       // contiguous range of every type's Gram matrix (one all-reduce of the partial sums)
    hipStream_t cs = nullptr;              // the stream the exchanges go out on
    std::vector<hipEvent_t> ev_own;
    bool masters_stale = false;            // SKF_BF16: rows of other owners hold old f32 values until finalize_owned
    // row-block sharding: contiguous ranges of all W, of the Q of unmasked / of masked relations
    bool sliced = false;
    size_t xw_off = 0, xw_bytes = 0, xq_off = 0, xq_bytes = 0, xqm_off = 0, xqm_bytes = 0;
    size_t btot_off = 0, btot_bytes = 0;   // contiguous range of every type's Bp_tot / Bn_tot
    void* ws_base = nullptr;
    // one captured iteration (hipGraph) for launch-bound graphs; replayed by skf_iterate
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t graph_stream = nullptr;
    bool graph_failed = false;
    bool graph_on = false;                 // skf_plan_set_graph
    // optional hipEvent timing of the relation contractions (skf_plan_set_profiling)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double prof_flops = 0.0, prof_bytes = 0.0;
    int64_t prof_launches = 0;
    bool kn_first = true;                  // no residuals stored yet: E = the known entries themselves, S_prev = 0
    // small graphs (skf_small.h): the whole DFMF iteration as eight launches over job tables kept in the workspace
    bool small_fused = false;
    skf::Slot sm_tables, sm_jobs1, sm_jobs3, sm_wpart, sm_gpart, sm_tickets, sm_batch;
    std::vector<const void*> sm_batch_host;        // device addresses of the tables of the plans of the last batch (this plan first)
    std::vector<skf::SmJob> sm_j1, sm_j3;
    ~skf_plan() {
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_rel) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_own) (void)hipEventDestroy(e);
        if (cs) (void)hipStreamDestroy(cs);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (aux) (void)hipStreamDestroy(aux);
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    }
};

namespace skf {

static void add_slot(skf_plan* p, Slot& s, size_t bytes) {
    s.bytes = bytes;
    s.off = p->ws_bytes;
    p->ws_bytes += align_up(bytes ? bytes : 1, 256);
    p->slots.push_back(&s);
}

static hipStream_t as_stream(void* s) { return (hipStream_t)s; }

static void copy2d(void* dst, int64_t ldd, const void* src, int64_t lds, int64_t rows, int64_t cols,
                   size_t esz, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return;
    SKF_HIP(hipMemcpy2DAsync(dst, (size_t)ldd * esz, src, (size_t)lds * esz, (size_t)cols * esz, (size_t)rows,
                             hipMemcpyDeviceToDevice, st));
}

static GemmArgs gemm_args(const void* A, int64_t sa_m, int64_t sa_k, const void* B, int64_t sb_k, int64_t sb_n,
                          void* C, int64_t ldc, int M, int N, int K, int epi, int nan) {
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = A; g.B = B; g.C = C; g.C2 = nullptr; g.mask = nullptr; g.part = nullptr;
    g.sa_m = sa_m; g.sa_k = sa_k; g.sb_k = sb_k; g.sb_n = sb_n;
    g.ldc = ldc; g.ldc2 = ldc; g.ldmask = 0;
    g.M = M; g.N = N; g.K = K; g.k_chunk = K;
    g.aop = AOP_NONE; g.epi = epi; g.nan_to_num = nan;
    return g;
}

// master x master -> master
static void plan_gemm(skf_plan* p, GemmArgs g, hipStream_t st) {
    run_gemm(GemmTypes{p->mt, p->mt, p->mt}, p->engine, g, 0, p->part.ptr, p->part_bytes, st);
}
// c x c algebra: f64 x f64 -> f64.  A launch on the second stream takes that stream's split-K scratch (ranks above 512 leave
// the deep unsplit tile and are cut into K slices: the main stream's partials may be in flight in `part` at that moment)
static void small_gemm(skf_plan* p, GemmArgs g, hipStream_t st) {
    const bool on_aux = p->aux != nullptr && st == p->aux;
    run_gemm(GemmTypes{SKF_F64, SKF_F64, SKF_F64}, p->engine, g, 0, on_aux ? p->part_aux.ptr : p->part.ptr,
             on_aux ? p->part_aux_bytes : p->part_bytes, st);
}
// master x master -> f64 (Gram, G^T P: long f64 accumulation over the object dimension)
static void wide_gemm(skf_plan* p, GemmArgs g, hipStream_t st) {
    run_gemm(GemmTypes{SKF_F64, p->mt, p->mt}, p->engine, g, 0, p->part.ptr, p->part_bytes, st);
}
// master x f64 -> master (n x c x c products with an f64 backbone / B / D matrix)
static void mixed_gemm(skf_plan* p, GemmArgs g, hipStream_t st) {
    run_gemm(GemmTypes{p->mt, p->mt, SKF_F64}, p->engine, g, 0, p->part.ptr, p->part_bytes, st);
}

// the same on the second stream: one K slice (the split-K scratch belongs to the main stream)
static void mixed_gemm_unsplit(skf_plan* p, GemmArgs g, hipStream_t st) {
    run_gemm(GemmTypes{p->mt, p->mt, SKF_F64}, p->engine, g, 1, nullptr, 0, st);
}

static hipEvent_t next_event(skf_plan* p) {
    if (p->ev_used == p->ev_pool.size()) {
        hipEvent_t e;
        SKF_HIP(hipEventCreate(&e));
        p->ev_pool.push_back(e);
    }
    return p->ev_pool[p->ev_used++];
}

// one of the two contractions that stream a relation matrix: P = R G_j or Q = R^T G_i
static void relation_gemm(skf_plan* p, GemmArgs g, hipStream_t st, const RelState* r = nullptr, bool is_q = false) {
    if (p->profiling) SKF_HIP(hipEventRecord(next_event(p), st));
    if (p->bf16) {
        const TypeState& ti = p->types[r->row];
        const TypeState& tj = p->types[r->col];
        if (r->sparse && r->sp_gather) {       // the ones as lists over the bf16 rows of the factor, every entry counting 1
            const TypeState& tin = is_q ? ti : tj;           // the gathered factor: Q = R^T G_i sums rows of G_i, P = R G_j rows of G_j
            const int parts = is_q ? r->sp_pr : r->sp_pc;
            const int64_t in0 = is_q ? r->r0 : 0, n_in = tin.n - in0;
            SrpArgs<uint16_t, float> a;
            memset(&a, 0, sizeof a);
            a.ptr = (const int64_t*)(parts > 1 ? (is_q ? r->SpCpP.ptr : r->SpRpP.ptr) : (is_q ? r->SpCp.ptr : r->SpRp.ptr));
            a.idx = (const int*)(is_q ? r->SpRi.ptr : r->SpCi.ptr);
            a.Fi = (const uint16_t*)tin.Grow.ptr + in0 * tin.ldrow;
            a.ldi = tin.ldrow; a.ldo = tin.ldrow;
            a.n_out = g.M; a.w = g.N; a.parts = parts; a.mode = SRP_ONES;
            a.ld_out = parts > 1 ? g.N : g.ldc;
            a.part_stride = (int64_t)g.M * g.N;
            a.out = parts > 1 ? (float*)p->sp_part.ptr : (float*)g.C;
            a.zero_off = (uint32_t)(n_in * tin.ldrow * 2);
            if (((int64_t)(n_in + 1) * tin.ldrow * 2) >= (int64_t)0xffffffffLL || tin.ldrow != g.N)
                SKF_FAIL(SKF_E_STATE, "sparse 0/1 relation: the gathered factor does not fit the list kernel");
            const int per = 8 / parts;
            const int64_t wgs = ((int64_t)g.M + 3) / 4;
            const int grid = (int)((wgs + per - 1) / per * 8);
            if (g.N == 64) hipLaunchKernelGGL((srp_bf16_v6_kernel<1, SRP_ONES, 2>), dim3(grid), dim3(256), 0, st, a);
            else if (g.N == 128) hipLaunchKernelGGL((srp_bf16_v6_kernel<1, SRP_ONES, 4>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((srp_bf16_v6_kernel<2, SRP_ONES, 4>), dim3(grid), dim3(256), 0, st, a);
            check_launch("sparse 0/1 relation (lists)");
            if (parts > 1) {
                if (g.ldc == g.N) {
                    const int64_t total = (int64_t)g.M * g.N;
                    hipLaunchKernelGGL((sum_parts_kernel<float>), dim3(elem_grid(total)), dim3(256), 0, st, (float*)g.C,
                                       (const float*)p->sp_part.ptr, total, parts, total);
                    check_launch("sum_parts");
                } else {
                    SKF_FAIL(SKF_E_STATE, "sparse 0/1 relation: strided output with parted lists");
                }
            }
        } else if (r->sparse) {        // a handful of ones per row / column: gather the factor's f32 rows (binary_spmm_kernel)
            const int wgrid = (int)(((int64_t)g.M + 3) / 4 < 4096 ? ((int64_t)g.M + 3) / 4 : 4096);
            hipLaunchKernelGGL(binary_spmm_kernel, dim3(wgrid), dim3(256), 0, st,
                               (const int64_t*)(is_q ? r->SpCp.ptr : r->SpRp.ptr), (const int*)(is_q ? r->SpRi.ptr : r->SpCi.ptr),
                               (const float*)g.B, (int64_t)g.N, (float*)g.C, (int64_t)g.ldc, (int64_t)g.M, g.N);
            check_launch("binary_spmm");
        } else if (r->binary) {        // the relation as a bitmap: 1/16 of the bytes, expanded to bf16 0 / 1 on the way into LDS
            if (!is_q)
                run_gemm_bf16((const uint16_t*)r->Bb.ptr, r->ldbb, (const uint16_t*)tj.GTb.ptr, tj.ldgt, (float*)g.C,
                              g.ldc, g.M, g.N, (int)r->ldrb, 0, p->part.ptr, p->part_bytes, true, st, false, true);
            else
                run_gemm_bf16((const uint16_t*)r->Bb.ptr, r->ldbb, (const uint16_t*)ti.GTb.ptr + r->r0, ti.ldgt, (float*)g.C,
                              g.ldc, g.M, g.N, (int)r->kq, 0, p->part.ptr, p->part_bytes, true, st, true, true);
        } else if (!is_q)      // P = R G_j :  A = R (bf16), Bt = G_j^T (bf16)
            run_gemm_bf16((const uint16_t*)r->Rb.ptr, r->ldrb, (const uint16_t*)tj.GTb.ptr, tj.ldgt, (float*)g.C,
                          g.ldc, g.M, g.N, (int)r->ldrb, 0, p->part.ptr, p->part_bytes, true, st);
        else            // Q = R^T G_i :  A = the same row-major R read transposed out of LDS, Bt = G_i^T (bf16)
            run_gemm_bf16((const uint16_t*)r->Rb.ptr, r->ldrb, (const uint16_t*)ti.GTb.ptr + r->r0, ti.ldgt, (float*)g.C,
                          g.ldc, g.M, g.N, (int)r->kq, 0, p->part.ptr, p->part_bytes, true, st, true);
    } else {
        run_gemm(GemmTypes{p->mt, p->mt, p->mt}, p->engine, g, 0, p->part.ptr, p->part_bytes, st, true);
    }
    if (p->profiling) {
        SKF_HIP(hipEventRecord(next_event(p), st));
        // what the launch executes and the relation bytes it reads as stored (include/skfusion_hip.h, skf_plan_get_profile)
        const double cells = (double)g.M * (double)g.K;
        if (p->bf16 && r && r->sparse) {
            p->prof_flops += 2.0 * (double)r->sp_nnz * (double)g.N;
            p->prof_bytes += 4.0 * (double)r->sp_nnz;
        } else {
            p->prof_flops += 2.0 * cells * (double)g.N;
            p->prof_bytes += (p->bf16 && r && r->binary) ? cells / 8.0 : cells * (p->bf16 ? 2.0 : (double)p->esz);
        }
        p->prof_launches += 1;
    }
}

// SKF_BF16: refresh the bf16 transpose of a factor after it changed
static void refresh_gt(skf_plan* p, TypeState& t, hipStream_t st) {
    if (!p->bf16) return;
    launch_to_bf16<float>((uint16_t*)t.GTb.ptr, t.ldgt, (const float*)t.G.ptr, (int64_t)t.c, t.n, (int64_t)t.c, true, st);
    if (t.Grow.ptr)        // the bf16 rows for the list passes (their all-zero last row is never written)
        launch_to_bf16<float>((uint16_t*)t.Grow.ptr, t.ldrow, (const float*)t.G.ptr, (int64_t)t.c, t.n, (int64_t)t.c, false, st);
}

// Relative pivot threshold of the Cholesky fast path: below it the Gram matrix goes to the deflation / the
// eigen-solver.  SKF_PINV_JACOBI=1 forces the eigen path with its exact singular-value cut-off (tests).
static double chol_rel_threshold(const Switches& sw) { return sw.pinv_jacobi ? 1e300 : 1e-8; }
// lower edge of the deflation's gap test
static double deflation_lo(const Switches& sw) { return sw.pinv_jacobi ? 1e300 : 1e-10; }

// Cholesky fast path: the LDS-blocked kernel up to order CHOLB_MAXN, the plain one beyond
static void launch_chol(const Switches& sw, const EighArgs& e, int batch, int max_order, hipStream_t st) {
    if (max_order <= CHOLS_MAXN && !sw.chol_no_small && !sw.chol_unblocked) {
        hipLaunchKernelGGL(chol_inverse_small_kernel, dim3((unsigned)batch), dim3(64), 0, st, e, chol_rel_threshold(sw));
    } else if (max_order <= CHOLB_MAXN && !sw.chol_unblocked) {
        size_t wave_tiles = (size_t)(EIGH_THREADS / 64) * CHOLB_NB * (CHOLB_NB + 1);
        size_t panel = (size_t)CHOLB_NB * max_order;
        size_t smem = ((size_t)CHOLB_NB * (CHOLB_NB + 1) + (panel > wave_tiles ? panel : wave_tiles)) * sizeof(double);
        static DeviceOnce once;
        allow_dynamic_lds(once, chol_inverse_blocked_kernel,
                          (int)(((size_t)CHOLB_NB * (CHOLB_NB + 1) + (size_t)CHOLB_NB * CHOLB_MAXN) * sizeof(double)));
        hipLaunchKernelGGL(chol_inverse_blocked_kernel, dim3((unsigned)batch), dim3(EIGH_THREADS), smem, st, e,
                           chol_rel_threshold(sw));
    } else {
        hipLaunchKernelGGL(chol_inverse_kernel, dim3((unsigned)batch), dim3(EIGH_THREADS), 0, st, e, chol_rel_threshold(sw));
    }
    check_launch("chol_inverse");
}

// The blocked sweep of `nb` matrices of order <= max_c: from order sw.sweep_step_min (default: always) one launch per block
// step with the rank-32 update of the step spread over row slabs -- orders up to SWEEP_MAXN with the panel in LDS, up to
// EIGH_MAXN with the column operands from memory --, else (orders <= SWEEP_MAXN) one workgroup per matrix in one launch.
static bool sweep_steps(const Switches& sw, int max_c) { return sw.sweep_step_min > 0 && max_c >= sw.sweep_step_min; }
static bool sweep_takes(const Switches& sw, int max_c) {
    if (max_c <= CHOLS_MAXN || sw.no_sweep || sw.chol_unblocked || sw.pinv_jacobi) return false;
    return max_c <= SWEEP_MAXN || (sweep_steps(sw, max_c) && max_c <= EIGH_MAXN && !sw.no_sweep_big);
}
static void launch_sweep(const Switches& sw, const EighArgs& e, const PinvBatch& pb, int nb, int max_c, hipStream_t st) {
    if (sweep_steps(sw, max_c)) {
        const bool big = max_c > SWEEP_MAXN;
        static DeviceOnce once, once_big;
        if (big) allow_dynamic_lds(once_big, sweep_step_kernel<true>, SWEEP_BIG_LDS_BYTES);
        else allow_dynamic_lds(once, sweep_step_kernel<false>, SWEEP_LDS_BYTES);
        const int rs = big ? SWEEP_NB : sw.sweep_rows;
        const int slabs = (max_c + rs - 1) / rs, steps = (max_c + SWEEP_NB - 1) / SWEEP_NB;
        for (int step = 0; step < steps; ++step) {
            if (big)
                hipLaunchKernelGGL(sweep_step_kernel<true>, dim3((unsigned)nb, (unsigned)slabs), dim3(SWEEP_THREADS), SWEEP_BIG_LDS_BYTES, st,
                                   e, pb, chol_rel_threshold(sw), step, rs);
            else
                hipLaunchKernelGGL(sweep_step_kernel<false>, dim3((unsigned)nb, (unsigned)slabs), dim3(SWEEP_THREADS), SWEEP_LDS_BYTES, st, e,
                                   pb, chol_rel_threshold(sw), step, rs);
            check_launch("sweep_step");
        }
        return;
    }
    static DeviceOnce once;
    allow_dynamic_lds(once, sweep_inverse_kernel, SWEEP_LDS_BYTES);
    hipLaunchKernelGGL(sweep_inverse_kernel, dim3((unsigned)nb), dim3(SWEEP_THREADS), SWEEP_LDS_BYTES, st, e, pb, chol_rel_threshold(sw));
    check_launch("sweep_inverse");
}

// Rank-revealing deflation over several workgroups (skf_kernels.h, pchol_step_kernel) for the matrices of a batch whose fast
// path failed, orders above SWEEP_MAXN: every launch is gated on the device (chol_ok, the verdict of the steps, the verdict of
// the sweep over B), the host issues the sequence blind.  K[b]: c x c f64, ld = c.  `scratch`: defl_scratch_bytes(nb, stride).
static size_t defl_scratch_bytes(int nb, int64_t stride) {
    return align_up((size_t)3 * nb * stride * 8, 256) + align_up((size_t)nb * 2 * EIGH_MAXN * 8, 256) +
           align_up((size_t)nb * 4 * 8, 256) + align_up((size_t)nb * 8 * sizeof(int), 256);
}
static bool defl_multi_takes(const Switches& sw, int max_c) {
    return max_c > SWEEP_MAXN && max_c < EIGH_MAXN && sweep_takes(sw, max_c) && !sw.pinv_jacobi;
}
static void launch_deflation_multi(const Switches& sw, int engine, const EighArgs& e, const PinvBatch& pb, int nb, int max_c,
                                   void* scratch, hipStream_t st) {
    char* base = (char*)scratch;
    double* M0 = (double*)base;
    double* M1 = M0 + (size_t)nb * e.stride;
    double* Binv = M1 + (size_t)nb * e.stride;
    base += align_up((size_t)3 * nb * e.stride * 8, 256);
    DeflArgs da;
    da.d = (double*)base;
    base += align_up((size_t)nb * 2 * EIGH_MAXN * 8, 256);
    da.vals = (double*)base;
    base += align_up((size_t)nb * 4 * 8, 256);
    da.state = (int*)base;
    da.n_defl = da.state + 4 * nb;
    da.gate = da.n_defl + nb;
    da.ok2 = da.gate + nb;
    da.rank = da.ok2 + nb;
    const int np = max_c + (max_c & 1);
    hipLaunchKernelGGL(pchol_init_kernel, dim3(elem_grid((int64_t)np * np), (unsigned)nb), dim3(256), 0, st, e, da);
    check_launch("pchol_init");
    const int steps = 2 * cdiv(max_c, DEFL_NB) + 1, slabs = cdiv(max_c, DEFL_ROWS);
    for (int step = 0; step < steps; ++step) {
        hipLaunchKernelGGL(pchol_step_kernel, dim3((unsigned)nb, (unsigned)slabs), dim3(DEFL_THREADS), 0, st, e, da, deflation_lo(sw), 1e-7,
                           step);
        check_launch("pchol_step");
    }
    const int fin = steps & 1;
    hipLaunchKernelGGL(pchol_verdict_kernel, dim3((unsigned)nb), dim3(64), 0, st, e, da, fin);
    check_launch("pchol_verdict");
    const GemmTypes f64{SKF_F64, SKF_F64, SKF_F64};
    for (int b = 0; b < nb; ++b) {                      // B = L^T L  (Lt[k][i] = L[i][k], ld = the padded order)
        const int c = pb.c[b], ld = pb.n_pad[b];
        const double* Lt = e.V + (int64_t)b * e.stride;
        GemmArgs g = gemm_args(Lt, ld, 1, Lt, 1, ld, e.Vs + (int64_t)b * e.stride, ld, c, c, c, EPI_STORE, 0);
        g.gate = da.gate + b;
        run_gemm(f64, engine, g, 1, nullptr, 0, st);
    }
    hipLaunchKernelGGL(pchol_patch_kernel, dim3((unsigned)nb), dim3(256), 0, st, e, da, fin);
    check_launch("pchol_patch");
    {                                                   // B^-1 by the blocked sweep of the fast path (idle where n_defl = 0)
        EighArgs e2 = e;
        e2.A = e.Vs; e2.V = M0; e2.Vs = M1;
        e2.n_orig = da.n_defl;
        e2.chol_ok = da.ok2;
        PinvBatch pb2 = pb;
        for (int b = 0; b < nb; ++b) pb2.K[b] = Binv + (int64_t)b * e.stride;
        launch_sweep(sw, e2, pb2, nb, max_c, st);
    }
    hipLaunchKernelGGL(pchol_gate2_kernel, dim3((unsigned)nb), dim3(64), 0, st, da);
    check_launch("pchol_gate2");
    for (int b = 0; b < nb; ++b) {
        const int c = pb.c[b], ld = pb.n_pad[b];
        const double* Lt = e.V + (int64_t)b * e.stride;
        const double* Bi = Binv + (int64_t)b * e.stride;        // (the sweep writes its result with ld = the order)
        double* Yt = M0 + (int64_t)b * e.stride;
        GemmArgs g = gemm_args(Bi, c, 1, Lt, ld, 1, Yt, ld, c, c, c, EPI_STORE, 0);         // Y^T = B^-1 L^T
        g.gate = da.gate + b;
        run_gemm(f64, engine, g, 1, nullptr, 0, st);
        g = gemm_args(Yt, 1, ld, Yt, ld, 1, pb.K[b], c, c, c, c, EPI_STORE, 0);             // K = Y Y^T
        g.gate = da.gate + b;
        run_gemm(f64, engine, g, 1, nullptr, 0, st);
    }
    hipLaunchKernelGGL(pchol_done_kernel, dim3((unsigned)nb), dim3(64), 0, st, e, da);
    check_launch("pchol_done");
}

// K_i = pinv(Gram_i) for every type (one workgroup each); `which` = 0..n_types-1, the order
// of the per-matrix order arrays uploaded once by skf_plan_bind_workspace.
static void pinv_fallbacks(skf_plan* p, const std::vector<int>& which, const PinvBatch& pb, const EighArgs& e, bool batched, int max_c,
                           hipStream_t st);
static void plan_pinv(skf_plan* p, const std::vector<int>& which, hipStream_t st) {
    if (which.empty()) return;
    const int64_t stride = p->eig_stride;
    const int nb = (int)which.size();
    const bool batched = nb <= PINV_MAXB;       // one launch for all types (pointers travel as kernel arguments)
    PinvBatch pb;
    int max_pad = 2, max_c = 1;
    if (batched) {
        for (int b = 0; b < nb; ++b) {
            const TypeState& t = p->types[which[b]];
            pb.gram[b] = (const double*)t.Gram.ptr;
            pb.K[b] = (double*)t.K.ptr;
            pb.c[b] = t.c;
            pb.n_pad[b] = t.n_pad;
            if (t.n_pad > max_pad) max_pad = t.n_pad;
            if (t.c > max_c) max_c = t.c;
        }
        hipLaunchKernelGGL(eigh_pack_batched_kernel, dim3(elem_grid((int64_t)max_pad * max_pad), nb), dim3(256), 0, st, pb,
                           (double*)p->eigA.ptr, stride);
        check_launch("eigh_pack");
    } else {
        for (size_t b = 0; b < which.size(); ++b) {
            const TypeState& t = p->types[which[b]];
            double* A = (double*)p->eigA.ptr + (int64_t)b * stride;
            const int total = t.n_pad * t.n_pad;
            hipLaunchKernelGGL((eigh_pack_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, st, A, t.n_pad,
                               (const double*)t.Gram.ptr, (int64_t)t.c, t.c);
            check_launch("eigh_pack");
        }
    }
    EighArgs e;
    e.A = (double*)p->eigA.ptr; e.V = (double*)p->eigV.ptr; e.Vs = (double*)p->eigVs.ptr;
    e.w = (double*)p->eigW.ptr; e.stride = stride; e.wstride = p->eig_maxn;
    e.n = (const int*)p->eigN.ptr; e.n_orig = (const int*)p->eigNorig.ptr;
    e.chol_ok = (int*)p->eigOk.ptr;
    e.max_sweeps = 30;
    // fast path (Cholesky inverse) with an on-device verdict; the Jacobi eigen-solver only does
    // work for the matrices the fast path rejected -- no host round trip either way
    // orders 65 .. 256 (round 4): the blocked sweep operator writes K itself -- one launch instead of the Cholesky inverse and
    // its unpack (1.15 + 0.09 ms at order 256)
    const bool sweep = batched && sweep_takes(p->sw, max_c);
    if (sweep) {
        launch_sweep(p->sw, e, pb, nb, max_c, st);
        pinv_fallbacks(p, which, pb, e, batched, max_c, st);
        return;
    }
    launch_chol(p->sw, e, nb, p->eig_maxn, st);
    if (batched) {
        hipLaunchKernelGGL(chol_unpack_batched_kernel, dim3(elem_grid((int64_t)max_c * max_c), nb), dim3(256), 0, st, pb,
                           (const double*)p->eigV.ptr, stride, (const int*)p->eigOk.ptr);
        check_launch("chol_unpack");
    } else {
        for (size_t b = 0; b < which.size(); ++b) {
            const TypeState& t = p->types[which[b]];
            const double* X = (const double*)p->eigV.ptr + (int64_t)b * stride;
            hipLaunchKernelGGL((chol_unpack_kernel<double>), dim3(elem_grid(t.c * t.c)), dim3(256), 0, st,
                               (double*)t.K.ptr, (int64_t)t.c, X, t.n_pad, t.c, (const int*)p->eigOk.ptr + b);
            check_launch("chol_unpack");
        }
    }
    pinv_fallbacks(p, which, pb, e, batched, max_c, st);
}

// the matrices the Cholesky fast path declined (verdicts in eigOk, packed copies in eigA); no-ops for the others
static void pinv_fallbacks(skf_plan* p, const std::vector<int>& which, const PinvBatch& pb, const EighArgs& e, bool batched, int max_c,
                           hipStream_t st) {
    const int64_t stride = p->eig_stride;
    const int nb = (int)which.size();
    // orders above 256: the deflation over several workgroups first (round 6); what it declines is still there for the rest
    if (batched && p->eigX.ptr && p->engine == SKF_ENGINE_MFMA && defl_multi_takes(p->sw, max_c))
        launch_deflation_multi(p->sw, p->engine, e, pb, nb, max_c, p->eigX.ptr, st);
    // a rank-deficient Gram matrix with a clear spectral gap: rank-revealing deflation (pchol_pinv_kernel); what it
    // declines goes to the eigen-solver with the exact singular-value cut-off
    {
        static DeviceOnce once;
        allow_dynamic_lds(once, pchol_pinv_kernel, PCHOL_LDS_BYTES);
    }
    {   // dynamic LDS for the packed factor of L^T L, sized by the largest order of this plan (a no-op launch still has
        // to find a CU with that much LDS free, so small graphs reserve little)
        const int lr = p->eig_maxn < PCHOL_LDS_R ? p->eig_maxn : PCHOL_LDS_R;
        hipLaunchKernelGGL(pchol_pinv_kernel, dim3((unsigned)which.size()), dim3(EIGH_THREADS), (size_t)lr * (lr + 1) / 2 * 8, st, e,
                           deflation_lo(p->sw), 1e-7, lr);
    }
    check_launch("pchol_pinv");
    hipLaunchKernelGGL(jacobi_eigh_kernel, dim3((unsigned)which.size()), dim3(EIGH_THREADS), 0, st, e);
    check_launch("jacobi_eigh");
    if (batched) {
        hipLaunchKernelGGL(eigh_unpack_pinv_batched_kernel, dim3(elem_grid((int64_t)max_c * max_c), nb), dim3(256), 0, st, pb,
                           (const double*)p->eigVs.ptr, (const double*)p->eigV.ptr, stride, (const int*)p->eigOk.ptr);
        check_launch("eigh_unpack");
    } else {
        for (size_t b = 0; b < which.size(); ++b) {
            const TypeState& t = p->types[which[b]];
            const double* Vs = (const double*)p->eigVs.ptr + (int64_t)b * stride;
            const double* V = (const double*)p->eigV.ptr + (int64_t)b * stride;
            const int total = t.c * t.c;
            hipLaunchKernelGGL((eigh_unpack_pinv_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, st,
                               (double*)t.K.ptr, (int64_t)t.c, Vs, V, t.n_pad, t.c, (const int*)p->eigOk.ptr + b);
            check_launch("eigh_unpack");
        }
    }
}

static void gram(skf_plan* p, TypeState& t, int nan, hipStream_t st, bool on_aux = false) {
    // Gram = G^T G : A = G^T (m-contiguous), B = G; f64 accumulation
    GemmArgs g = gemm_args(t.G.ptr, 1, t.c, t.G.ptr, t.c, 1, t.Gram.ptr, t.c, t.c, t.c, (int)t.n, EPI_STORE, nan);
    g.sym = p->sw.gram_sym;
    run_gemm(GemmTypes{SKF_F64, p->mt, p->mt}, p->engine, g, 0, on_aux ? p->part_aux.ptr : p->part.ptr,
             on_aux ? p->part_aux_bytes : p->part_bytes, st);
}

static void mult_update(skf_plan* p, TypeState& t, hipStream_t st) {
    const int64_t total = t.n * t.c;
    if (p->f64)
        hipLaunchKernelGGL((mult_update_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, st, (double*)t.G.ptr,
                           (const double*)t.E.ptr, (const double*)t.D.ptr, t.n, t.c, (int64_t)t.c, (int64_t)t.c);
    else
        hipLaunchKernelGGL((mult_update_kernel<float>), dim3(elem_grid(total)), dim3(256), 0, st, (float*)t.G.ptr,
                           (const float*)t.E.ptr, (const float*)t.D.ptr, t.n, t.c, (int64_t)t.c, (int64_t)t.c);
    check_launch("mult_update");
}

// D_i += Theta+ G_i ; E_i += Theta- G_i   (_dfmf.py:284-292) on the rows [r0, r0 + nr) of the constrained type (all of
// them, or -- ownership-aligned row blocks -- the rows this plan owns: a row of the product needs the row of Theta and the
// whole factor).  `only_type` >= 0: the constraints of that type only.
static void theta_terms_rows(skf_plan* p, int only_type, bool own_rows, hipStream_t st) {
    for (ThetaState& th : p->thetas) {
        if (only_type >= 0 && th.type != only_type) continue;
        TypeState& t = p->types[th.type];
        const int64_t r0 = own_rows ? t.t0 : 0, nr = own_rows ? t.tn : t.n;
        if (nr <= 0) continue;
        void* Er = (char*)t.E.ptr + (size_t)r0 * t.c * p->esz;
        void* Dr = (char*)t.D.ptr + (size_t)r0 * t.c * p->esz;
        // an all-zero half is skipped
        if (th.sparse) {    // both halves in one pass over the CSR form, master precision
            if (th.nnz == 0) continue;
            const int grid = (int)((nr + 3) / 4 < 2048 ? (nr + 3) / 4 : 2048);
            if (p->f64)
                hipLaunchKernelGGL((theta_spmm_kernel<double>), dim3(grid), dim3(256), 0, st, (const int64_t*)th.Rp.ptr + r0,
                                   (const int*)th.Ci.ptr, (const double*)th.Vv.ptr, (const double*)t.G.ptr, (double*)Er,
                                   (double*)Dr, nr, t.c);
            else
                hipLaunchKernelGGL((theta_spmm_kernel<float>), dim3(grid), dim3(256), 0, st, (const int64_t*)th.Rp.ptr + r0,
                                   (const int*)th.Ci.ptr, (const float*)th.Vv.ptr, (const float*)t.G.ptr, (float*)Er,
                                   (float*)Dr, nr, t.c);
            check_launch("theta_spmm");
            continue;
        }
        if (p->bf16) {      // bf16 copies of the halves against the stored G^T, f32 accumulate, then added
            for (int half = 0; half < 2; ++half) {
                if (!(half == 0 ? th.has_pos : th.has_neg)) continue;
                run_gemm_bf16((const uint16_t*)(half == 0 ? th.Pb.ptr : th.Nb.ptr) + r0 * th.ldb, th.ldb, (const uint16_t*)t.GTb.ptr,
                              t.ldgt, (float*)p->theta_tmp.ptr, t.c, (int)nr, t.c, (int)th.ldb, 0, p->part.ptr,
                              p->part_bytes, false, st);
                hipLaunchKernelGGL(add_into_kernel, dim3(elem_grid(nr * t.c)), dim3(256), 0, st,
                                   (float*)(half == 0 ? Dr : Er), (const float*)p->theta_tmp.ptr, (int64_t)nr * t.c);
                check_launch("add_into");
            }
            continue;
        }
        GemmArgs g = gemm_args((const char*)th.data + (size_t)r0 * th.ld * p->esz, th.ld, 1, t.G.ptr, t.c, 1, Dr, t.c, (int)nr, t.c,
                               (int)t.n, EPI_ACC, 0);
        g.aop = AOP_POS;
        if (th.has_pos) plan_gemm(p, g, st);
        g.C = Er;
        g.aop = AOP_NEG;
        if (th.has_neg) plan_gemm(p, g, st);
    }
}

static void theta_terms(skf_plan* p, hipStream_t st) { theta_terms_rows(p, -1, false, st); }

// [Xp, Xn] (+)= split( L * Gram * Rr )  helper for B = S Gram_j S^T and D = S^T Gram_i S
// tmp = first product, then split-store/acc of the second.
static void relation_small_terms(skf_plan* p, RelState& r, int nan_upd, int epi_split, void* Bp, void* Bn,
                                 void* Dp, void* Dn, bool want_row, bool want_col, hipStream_t st, void* U2 = nullptr) {
    TypeState& ti = p->types[r.row];
    TypeState& tj = p->types[r.col];
    const int ci = ti.c, cj = tj.c;
    if (want_row && want_col && U2 && U2 != r.U.ptr) {
        // both sides, and a second ci x cj buffer: the two Gram products side by side in one launch, then the two products
        // that add into the B sums (one launch as well unless both add into the SAME sums: a relation of a type with itself)
        const bool pairs = !p->sw.no_pairs;
        GemmArgs gu = gemm_args(r.S.ptr, cj, 1, tj.Gram.ptr, cj, 1, r.U.ptr, cj, ci, cj, cj, EPI_STORE, 0);      // U = S Gram_j
        GemmArgs gv = gemm_args(ti.Gram.ptr, ci, 1, r.S.ptr, cj, 1, U2, cj, ci, cj, ci, EPI_STORE, 0);          // U2 = Gram_i S
        run_gemm_pair_f64(p->engine, gu, gv, st, pairs);
        GemmArgs gb = gemm_args(r.U.ptr, cj, 1, r.S.ptr, 1, cj, Bp, ci, ci, ci, cj, epi_split, nan_upd);         // B = U S^T
        gb.C2 = Bn;
        GemmArgs gd = gemm_args(r.S.ptr, 1, cj, U2, cj, 1, Dp, cj, cj, cj, ci, epi_split, nan_upd);              // D = S^T U2
        gd.C2 = Dn;
        run_gemm_pair_f64(p->engine, gb, gd, st, pairs && Bp != Dp && Bn != Dn);
        return;
    }
    if (want_row) {
        // U = S Gram_j (ci x cj);  B = U S^T (ci x ci)          tmp2 of _dfmf.py:260
        GemmArgs g = gemm_args(r.S.ptr, cj, 1, tj.Gram.ptr, cj, 1, r.U.ptr, cj, ci, cj, cj, EPI_STORE, 0);
        small_gemm(p, g, st);
        g = gemm_args(r.U.ptr, cj, 1, r.S.ptr, 1, cj, Bp, ci, ci, ci, cj, epi_split, nan_upd);
        g.C2 = Bn;
        small_gemm(p, g, st);
    }
    if (want_col) {
        // U = Gram_i S (ci x cj);  D = S^T U (cj x cj)          tmp5 of _dfmf.py:272
        GemmArgs g = gemm_args(ti.Gram.ptr, ci, 1, r.S.ptr, cj, 1, r.U.ptr, cj, ci, cj, ci, EPI_STORE, 0);
        small_gemm(p, g, st);
        g = gemm_args(r.S.ptr, 1, cj, r.U.ptr, cj, 1, Dp, cj, cj, cj, ci, epi_split, nan_upd);
        g.C2 = Dn;
        small_gemm(p, g, st);
    }
}

// f32 engines: the f32 roundings of a type's two B sums in ONE launch (they were two)
static void cast_b_sums(TypeState& t, hipStream_t st) {
    const int cc = t.c * t.c;
    CastBatch cb;
    memset(&cb, 0, sizeof cb);
    cb.src[0] = (const double*)t.Bn_tot.ptr; cb.dst[0] = (float*)t.Bn32.ptr; cb.count[0] = cc;
    cb.src[1] = (const double*)t.Bp_tot.ptr; cb.dst[1] = (float*)t.Bp32.ptr; cb.count[1] = cc;
    hipLaunchKernelGGL(cast_batched_kernel, dim3(elem_grid(cc), 2), dim3(256), 0, st, cb);
    check_launch("cast_batched");
}

// Fused E/D update of one relation side (MFMA engine): E (+)= (X Sop)+ + G Bn, D (+)= (X Sop)- + G Bp
// on the rows [G, E, D point at the first one; n of them] of type t.  Sop / Bn / Bp are c x c matrices
// in the master type (the f32 engines pass f32 roundings of the f64 originals -- the same values
// the f32 matrix cores would see -- so that the staged operands take half the registers).
static void side_update(skf_plan* p, const void* X, int64_t ldx, int k1, const void* Sop, int64_t ss_k, int64_t ss_n,
                        TypeState& t, const void* G, void* E, void* D, int n, const void* Bn, const void* Bp,
                        bool phase2, bool accumulate, int nan, hipStream_t st) {
    SideArgs a;
    a.X = X; a.Sop = Sop; a.G = G; a.Bn = Bn; a.Bp = Bp; a.E = E; a.D = D;
    a.ldx = ldx; a.ss_k = ss_k; a.ss_n = ss_n; a.ldg = t.c; a.ldb = t.c; a.lde = t.c;
    a.n = n; a.c = t.c; a.k1 = k1;
    a.accumulate = accumulate ? 1 : 0;
    a.nan_to_num = nan;
    a.phase2 = phase2 ? 1 : 0;
    bool big = (n > 64 && t.c > 64);
    dim3 block(GEMM_THREADS);
    if (p->f64) {
        if (big) {
            dim3 grid(cdiv(t.c, 64), cdiv(n, 64));
            hipLaunchKernelGGL((side_update_kernel<double, double, 2, 2, 16>), grid, block, 0, st, a);
        } else {
            dim3 grid(cdiv(t.c, 32), cdiv(n, 32));
            hipLaunchKernelGGL((side_update_kernel<double, double, 1, 1, 16>), grid, block, 0, st, a);
        }
    } else {
        const int force = p->sw.side_tile;                   // 64 / 128 force a tile shape (A/B runs, tests)
        // the two operand layouts of the iteration with everything 16-byte aligned get kernels whose
        // staging modes are compile-time constants (SKF_SIDE_FM); anything else the generic one
        auto al = [](const void* q) { return q == nullptr || (((uintptr_t)q) & 15) == 0; };
        const bool vec = al(X) && al(Sop) && al(G) && al(Bn) && al(Bp) && t.c % 4 == 0 && k1 % 4 == 0 &&
                         ldx % 4 == 0 && (k1 == 0 || (ss_k == 1 ? ss_n % 4 == 0 : (ss_n == 1 && ss_k % 4 == 0)));
        const bool k_major = (k1 == 0 || ss_k == 1);
        // measured at config 3 (rocprof, all six launches): the 64 x 64 fixed-mode kernels (78 VGPRs, 6
        // workgroups per CU) beat the 128 x 128 ones (182 VGPRs, 2 per CU): 0.98 vs 1.30 ms per iteration
        // (the 128 x 128 tile only with compile-time staging modes: its run-time form spilled registers)
        if (big && !(force == 128 && vec)) big = false;
        constexpr int FM_K = SKF_SIDE_FM(STAGE_VEC_K, STAGE_VEC_K, STAGE_VEC_K, STAGE_VEC_R);
        constexpr int FM_R = SKF_SIDE_FM(STAGE_VEC_K, STAGE_VEC_R, STAGE_VEC_K, STAGE_VEC_R);
        if (big) {
            dim3 grid(cdiv(t.c, 128), cdiv(n, 128));
            if (k_major)
                hipLaunchKernelGGL((side_update_kernel<float, float, 2, 2, 16, FM_K>), grid, block, 0, st, a);
            else
                hipLaunchKernelGGL((side_update_kernel<float, float, 2, 2, 16, FM_R>), grid, block, 0, st, a);
        } else {
            dim3 grid(cdiv(t.c, 64), cdiv(n, 64));
            // (K tiles of 32 / 64 instead of 16, round 4: config 3 91.1 / 89.2 against 90.8 it/s, config 5 144.9 / 143.0
            //  against 144.3 -- within the noise / worse)
            if (vec && k_major && n > 64 && t.c >= 64)
                hipLaunchKernelGGL((side_update_kernel<float, float, 1, 1, 16, FM_K>), grid, block, 0, st, a);
            else if (vec && n > 64 && t.c >= 64)
                hipLaunchKernelGGL((side_update_kernel<float, float, 1, 1, 16, FM_R>), grid, block, 0, st, a);
            else
                hipLaunchKernelGGL((side_update_kernel<float, float, 1, 1, 16>), grid, block, 0, st, a);
        }
    }
    check_launch("side_update");
}

// A view of the rows [r0, r0 + nr) of a factor-shaped matrix (G, E, D) of type t
static inline void* rows_of(const skf_plan* p, const Slot& s, const TypeState& t, int64_t r0) {
    return (char*)s.ptr + (size_t)r0 * t.c * p->esz;
}

static void contraction_P(skf_plan* p, RelState& r, hipStream_t st) {      // P = R_blk G_j
    TypeState& tj = p->types[r.col];
    GemmArgs g = gemm_args(r.R, r.ldr, 1, tj.G.ptr, tj.c, 1, r.P.ptr, tj.c, (int)r.nr, tj.c, (int)tj.n, EPI_STORE, 0);
    relation_gemm(p, g, st, &r, false);
}

static void contraction_Q(skf_plan* p, RelState& r, hipStream_t st) {      // Q = R_blk^T G_i[blk]
    TypeState& ti = p->types[r.row];
    TypeState& tj = p->types[r.col];
    if (r.absent) {
        SKF_HIP(hipMemsetAsync(r.Q.ptr, 0, r.Q.bytes, st));
        return;
    }
    GemmArgs g = gemm_args(r.R, 1, r.ldr, rows_of(p, ti.G, ti, r.r0), ti.c, 1, r.Q.ptr, ti.c, (int)tj.n, ti.c,
                           (int)r.nr, EPI_STORE, 0);
    relation_gemm(p, g, st, &r, true);
}

// ------------------------------------------------------------------------------------------
// DFMC on the known entries only (skf_known.h): launches of the list passes and the c x c / n x c x c algebra around them
// ------------------------------------------------------------------------------------------
template <typename TG, typename TM>
static int launch_srp(const SrpArgs<TG, TM>& a, hipStream_t st, bool generic) {
    const int per = 8 / a.parts;
    const int64_t wgs = (a.n_out + 3) / 4;
    const int grid = (int)((wgs + per - 1) / per * 8);
    constexpr int VE = 16 / (int)sizeof(TG);
    auto al = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    const bool vec = a.w % VE == 0 && a.ldi % VE == 0 && al(a.Fi) && (a.mode == SRP_APPLY || (a.ldo % VE == 0 && al(a.Fo)));
    const int gl = (vec && !generic) ? a.w / VE : 0;
    bool done = false;
    if constexpr (std::is_same<TG, uint16_t>::value) {
        // srp_bf16_v6_kernel: a row of 16 lanes per gathered vector, w = 128 (both modes) or 256 (SRP_APPLY: measured, the
        // residual pass at w = 256 is faster with two rows per vector), 32-bit byte offsets, an all-zero row behind Fi
        if (a.zero_off != 0 && a.mode != SRP_ERR && (gl == 16 || (gl == 32 && a.mode == SRP_APPLY))) {
            if (gl == 16 && a.mode == SRP_RESIDUAL) hipLaunchKernelGGL((srp_bf16_v6_kernel<1, SRP_RESIDUAL>), dim3(grid), dim3(256), 0, st, a);
            else if (gl == 16) hipLaunchKernelGGL((srp_bf16_v6_kernel<1, SRP_APPLY>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((srp_bf16_v6_kernel<2, SRP_APPLY>), dim3(grid), dim3(256), 0, st, a);
            check_launch("known-entry pass (v6)");
            return grid * 4;
        }
        done = true;
#define SKF_SRP_BF16(GL_)                                                                                                   \
    do {                                                                                                                    \
        if (a.mode == SRP_RESIDUAL) hipLaunchKernelGGL((srp_bf16_kernel<GL_, SRP_RESIDUAL>), dim3(grid), dim3(256), 0, st, a); \
        else if (a.mode == SRP_APPLY) hipLaunchKernelGGL((srp_bf16_kernel<GL_, SRP_APPLY>), dim3(grid), dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((srp_bf16_kernel<GL_, SRP_ERR>), dim3(grid), dim3(256), 0, st, a);                            \
    } while (0)
        if (gl == 8) SKF_SRP_BF16(8);
        else if (gl == 16) SKF_SRP_BF16(16);
        else if (gl == 32) SKF_SRP_BF16(32);
        else done = false;
#undef SKF_SRP_BF16
    }
    if (!done) {
        if (gl == 8) hipLaunchKernelGGL((srp_vec_kernel<TG, TM, 8>), dim3(grid), dim3(256), 0, st, a);
        else if (gl == 16) hipLaunchKernelGGL((srp_vec_kernel<TG, TM, 16>), dim3(grid), dim3(256), 0, st, a);
        else if (gl == 32) hipLaunchKernelGGL((srp_vec_kernel<TG, TM, 32>), dim3(grid), dim3(256), 0, st, a);
        else if (gl == 64) hipLaunchKernelGGL((srp_vec_kernel<TG, TM, 64>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((srp_any_kernel<TG, TM>), dim3(grid), dim3(256), 0, st, a);
    }
    check_launch("known-entry pass");
    return grid * 4;                 // waves = error partials of an SRP_ERR pass
}

// One pass over the known entries of relation r: by_col == false walks the row lists (outer = rows, gathers the rows of
// T = G_j S^T), by_col == true the column lists (outer = columns, gathers the rows of G_i).  Results land in r.A / r.Q
// (SRP_ERR: partials in p->sqpart, the number of which is returned).
static int known_pass(skf_plan* p, RelState& r, bool by_col, int mode, hipStream_t st, int sq_first = 0) {
    TypeState& ti = p->types[r.row];
    TypeState& tj = p->types[r.col];
    const int ci = ti.c;
    const int64_t n_out = by_col ? tj.n : r.nr;
    const int parts = by_col ? r.kn_pr : r.kn_pc;
    if (mode == SRP_ERR) {          // one error partial per wave, written from slot `sq_first` of p->sqpart on: room for all of them?
        const int64_t need = ((n_out + 3) / 4 + (8 / parts) - 1) / (8 / parts) * 8 * 4 + sq_first;
        if ((size_t)need > p->sq_elems) SKF_FAIL(SKF_E_STATE, "residual partials: %lld > %zu slots", (long long)need, p->sq_elems);
    }
    void* final_out = by_col ? r.Q.ptr : r.A.ptr;
    void* out = parts > 1 ? (by_col ? r.Qpart.ptr : r.Apart.ptr) : final_out;
    // vectors of the row objects of the block (bf16: kept in step with G^T); rows [r0, r0 + nr) of the type under row ownership
    const void* Gi = p->bf16 ? (const void*)((const char*)ti.Grow.ptr + (size_t)r.r0 * ti.ldrow * 2)
                             : (const void*)((const char*)ti.G.ptr + (size_t)r.r0 * ci * p->esz);
    const void* Tj = p->bf16 ? r.FiB.ptr : r.Tm.ptr;          // vectors of the column objects
    const int64_t ldv = p->bf16 ? r.kn_ldf : ci;
    if (p->profiling) SKF_HIP(hipEventRecord(next_event(p), st));
    int waves = 0;
    auto fill = [&](auto& a) {
        memset(&a, 0, sizeof a);
        a.ptr = (const int64_t*)(by_col ? r.KcPtr.ptr : r.KrPtr.ptr);
        a.idx = (const int*)(by_col ? r.KcIdx.ptr : r.KrIdx.ptr);
        a.ldo = ldv; a.ldi = ldv; a.ld_out = ci; a.part_stride = n_out * ci; a.n_out = n_out;
        a.w = ci; a.parts = parts; a.mode = mode;
        a.sq = (double*)p->sqpart.ptr + sq_first;
    };
    if (p->f64) {
        SrpArgs<double, double> a;
        fill(a);
        a.rvals = (const double*)(by_col ? r.KcVal.ptr : r.KrVal.ptr);
        a.evals = by_col ? (double*)r.KcE.ptr : nullptr;
        a.Fo = (const double*)(by_col ? Tj : Gi); a.Fi = (const double*)(by_col ? Gi : Tj);
        a.out = (double*)out;
        waves = launch_srp(a, st, false);
    } else if (p->bf16) {
        SrpArgs<uint16_t, float> a;
        fill(a);
        a.rvals = (const float*)(by_col ? r.KcVal.ptr : r.KrVal.ptr);
        a.evals = by_col ? (float*)r.KcE.ptr : nullptr;
        a.Fo = (const uint16_t*)(by_col ? Tj : Gi); a.Fi = (const uint16_t*)(by_col ? Gi : Tj);
        a.out = (float*)out;
        // the all-zero row behind the gathered matrix (TypeState::Grow / FiB hold one row more than the factor): slots past the end of
        // a list point there.  Byte offsets into the matrix are 32 bits wide in the v6 kernel.
        const int64_t n_in = by_col ? ti.n - r.r0 : tj.n;          // (the zero row of Grow sits behind ALL rows of the type)
        const int64_t zoff = n_in * ldv * 2;
        a.zero_off = (zoff + ldv * 2 < (int64_t)0xffffffffLL) ? (uint32_t)zoff : 0u;
        waves = launch_srp(a, st, false);
    } else {
        SrpArgs<float, float> a;
        fill(a);
        a.rvals = (const float*)(by_col ? r.KcVal.ptr : r.KrVal.ptr);
        a.evals = by_col ? (float*)r.KcE.ptr : nullptr;
        a.Fo = (const float*)(by_col ? Tj : Gi); a.Fi = (const float*)(by_col ? Gi : Tj);
        a.out = (float*)out;
        waves = launch_srp(a, st, false);
    }
    if (parts > 1 && mode != SRP_ERR) {
        const int64_t total = n_out * ci;
        if (p->f64)
            hipLaunchKernelGGL((sum_parts_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, st, (double*)final_out,
                               (const double*)out, total, parts, total);
        else
            hipLaunchKernelGGL((sum_parts_kernel<float>), dim3(elem_grid(total)), dim3(256), 0, st, (float*)final_out,
                               (const float*)out, total, parts, total);
        check_launch("sum_parts");
    }
    if (p->profiling) {
        SKF_HIP(hipEventRecord(next_event(p), st));
        const size_t gsz = p->bf16 ? 2 : p->esz;
        p->prof_flops += (mode == SRP_APPLY ? 2.0 : 4.0) * (double)r.kn_nnz * ci;
        p->prof_bytes += (double)r.kn_nnz * (4.0 + 2.0 * p->esz);          // index + value (+ residual) lists
        (void)gsz;
        p->prof_launches += 1;
    }
    return waves;
}


// W = G_i^T R_c G_j for the backbone (_dfmc.py:311-314) with R_c = G_i,prev S_prev G_j,prev^T + E_prev:
//     W = (G_i^T G_i,prev) S_prev (G_j,prev^T G_j) + (E_prev^T G_i)^T G_j
// first iteration: R_c = the known entries, zeros elsewhere (_dfmc.py:287-292), i.e. E_prev = R on the lists, S_prev = 0
// the two cross-Gram matrices of W: Xi = G_i^T G_i,prev, Xj = G_j,prev^T G_j (f64 accumulation over the objects)
static void known_cross(skf_plan* p, RelState& r, hipStream_t st, bool on_aux) {
    if (p->kn_first) return;
    TypeState& ti = p->types[r.row];
    TypeState& tj = p->types[r.col];
    const int ci = ti.c, cj = tj.c;
    void* part = on_aux ? p->part_aux.ptr : p->part.ptr;
    const size_t part_bytes = on_aux ? p->part_aux_bytes : p->part_bytes;
    // (over the rows of the block: under row ownership the block's own share of W, summed over the processes with the rest)
    GemmArgs g = gemm_args(rows_of(p, ti.G, ti, r.r0), 1, ci, rows_of(p, ti.Gp, ti, r.r0), ci, 1, r.Xi.ptr, ci, ci, ci, (int)r.nr,
                           EPI_STORE, 0);                                                                         // Xi = G_i^T G_i,prev
    run_gemm(GemmTypes{SKF_F64, p->mt, p->mt}, p->engine, g, 0, part, part_bytes, st);
    g = gemm_args(tj.Gp.ptr, 1, cj, tj.G.ptr, cj, 1, r.Xj.ptr, cj, cj, cj, (int)tj.n, EPI_STORE, 0);             // Xj = G_j,prev^T G_j
    run_gemm(GemmTypes{SKF_F64, p->mt, p->mt}, p->engine, g, 0, part, part_bytes, st);
}
static void known_w(skf_plan* p, RelState& r, hipStream_t st) {
    TypeState& ti = p->types[r.row];
    TypeState& tj = p->types[r.col];
    const int ci = ti.c, cj = tj.c;
    known_pass(p, r, true, SRP_APPLY, st);                                                          // Y = E_prev^T G_i  -> r.Q
    GemmArgs g = gemm_args(r.Q.ptr, 1, ci, tj.G.ptr, cj, 1, r.W.ptr, cj, ci, cj, (int)tj.n, EPI_STORE, 0);
    wide_gemm(p, g, st);                                                                // W = Y^T G_j
    if (p->kn_first) return;
    known_cross(p, r, st, false);
    g = gemm_args(r.Sp.ptr, cj, 1, r.Xj.ptr, cj, 1, r.U.ptr, cj, ci, cj, cj, EPI_STORE, 0);             // U = S_prev Xj
    small_gemm(p, g, st);
    g = gemm_args(r.Xi.ptr, ci, 1, r.U.ptr, cj, 1, r.W.ptr, cj, ci, cj, ci, EPI_ACC, 0);                // W += Xi U
    small_gemm(p, g, st);
}

// behind the backbone S: the gathered vectors T = G_j S^T and the c x c operands of the dense parts
//     U2 = S^T Gram_i (Q = G_j U2 + E^T G_i) ,  Bf = S Gram_j S^T (P S^T = G_i Bf + E T)
static void known_operands(skf_plan* p, RelState& r, hipStream_t st, bool second_stream) {
    TypeState& ti = p->types[r.row];
    TypeState& tj = p->types[r.col];
    const int ci = ti.c, cj = tj.c;
    GemmArgs g = gemm_args(tj.G.ptr, cj, 1, r.S.ptr, 1, cj, r.Tm.ptr, ci, (int)tj.n, ci, cj, EPI_STORE, 0);
    if (second_stream) mixed_gemm_unsplit(p, g, st);
    else mixed_gemm(p, g, st);
    if (p->bf16) launch_to_bf16<float>((uint16_t*)r.FiB.ptr, r.kn_ldf, (const float*)r.Tm.ptr, (int64_t)ci, tj.n, ci, false, st);
    g = gemm_args(r.S.ptr, 1, cj, ti.Gram.ptr, ci, 1, r.U2.ptr, ci, cj, ci, ci, EPI_STORE, 0);           // U2 = S^T Gram_i
    GemmArgs h = gemm_args(r.S.ptr, cj, 1, tj.Gram.ptr, cj, 1, r.T1.ptr, cj, ci, cj, cj, EPI_STORE, 0);    // T1 = S Gram_j
    run_gemm_pair_f64(p->engine, g, h, st, !p->sw.no_pairs);                                            // (independent: one launch)
    g = gemm_args(r.T1.ptr, cj, 1, r.S.ptr, 1, cj, r.Bf.ptr, ci, ci, ci, cj, EPI_STORE, 0);
    small_gemm(p, g, st);
}

// the two residual passes of the factor update: r.A = E T (row lists), r.Q = E^T G_i + G_j U2 (column lists; the
// residuals of this iteration replace the stored ones), and S_prev <- S
static void known_row_pass(skf_plan* p, RelState& r, hipStream_t st) { known_pass(p, r, false, SRP_RESIDUAL, st); }
// the dense part of Q on the rows [j0, j0 + nj) of the column type: Q += G_j (S^T Gram_i)
static void known_col_dense(skf_plan* p, RelState& r, int64_t j0, int64_t nj, hipStream_t st, bool second_stream) {
    TypeState& ti = p->types[r.row];
    TypeState& tj = p->types[r.col];
    const int ci = ti.c, cj = tj.c;
    if (nj <= 0) return;
    GemmArgs g = gemm_args(rows_of(p, tj.G, tj, j0), cj, 1, r.U2.ptr, ci, 1, (char*)r.Q.ptr + (size_t)j0 * ci * p->esz, ci, (int)nj, ci,
                           cj, EPI_ACC, 0);
    if (second_stream) mixed_gemm_unsplit(p, g, st);
    else mixed_gemm(p, g, st);
}
static void known_col_pass(skf_plan* p, RelState& r, hipStream_t st) {
    TypeState& ti = p->types[r.row];
    TypeState& tj = p->types[r.col];
    known_pass(p, r, true, SRP_RESIDUAL, st);
    known_col_dense(p, r, 0, tj.n, st, false);
    copy2d(r.Sp.ptr, tj.c, r.S.ptr, tj.c, ti.c, tj.c, 8, st);
}

// row side of the factor update from the row-side product itself: A = G_i Bf + E T, E_i (+)= A+, D_i (+)= A-
static void known_row_dense(skf_plan* p, RelState& r, hipStream_t st, bool second_stream) {
    TypeState& ti = p->types[r.row];
    const int ci = ti.c;
    GemmArgs g = gemm_args(rows_of(p, ti.G, ti, r.r0), ci, 1, r.Bf.ptr, ci, 1, r.A.ptr, ci, (int)r.nr, ci, ci, EPI_ACC, 0);
    if (second_stream) mixed_gemm_unsplit(p, g, st);
    else mixed_gemm(p, g, st);
}
static void known_row_split(skf_plan* p, RelState& r, bool accumulate, hipStream_t st) {
    TypeState& ti = p->types[r.row];
    const int ci = ti.c;
    const int64_t total = r.nr * ci;
    if (p->f64)
        hipLaunchKernelGGL((split_accumulate_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, st,
                           (double*)rows_of(p, ti.E, ti, r.r0), (double*)rows_of(p, ti.D, ti, r.r0), (const double*)r.A.ptr, total,
                           accumulate ? 1 : 0);
    else
        hipLaunchKernelGGL((split_accumulate_kernel<float>), dim3(elem_grid(total)), dim3(256), 0, st,
                           (float*)rows_of(p, ti.E, ti, r.r0), (float*)rows_of(p, ti.D, ti, r.r0), (const float*)r.A.ptr, total,
                           accumulate ? 1 : 0);
    check_launch("split_accumulate");
}
static void known_row_side(skf_plan* p, RelState& r, bool accumulate, hipStream_t st, bool second_stream) {
    known_row_dense(p, r, st, second_stream);
    known_row_split(p, r, accumulate, st);
}

// DFMC, first iteration: the unknown entries of every masked relation start at zero (_dfmc.py:287-292)
static void zero_unknown_entries(skf_plan* p, hipStream_t st) {
    for (RelState& r : p->rels) {
        if (!r.mask || r.kn) continue;          // (a known-entries relation starts from E = R on its lists, S_prev = 0)
        const int64_t rows = r.nr, cols = p->types[r.col].n;
        if (p->bf16)
            hipLaunchKernelGGL((mask_zero_kernel<uint16_t>), dim3(elem_grid(rows * cols)), dim3(256), 0, st,
                               (uint16_t*)r.Rb.ptr, r.ldrb, (const uint8_t*)r.Mb.ptr, r.ldmb, rows, cols);
        else if (p->f64)
            hipLaunchKernelGGL((mask_zero_kernel<double>), dim3(elem_grid(rows * cols)), dim3(256), 0, st,
                               (double*)r.Rw.ptr, r.ldr, (const uint8_t*)r.Mb.ptr, r.ldmb, rows, cols);
        else
            hipLaunchKernelGGL((mask_zero_kernel<float>), dim3(elem_grid(rows * cols)), dim3(256), 0, st,
                               (float*)r.Rw.ptr, r.ldr, (const uint8_t*)r.Mb.ptr, r.ldmb, rows, cols);
        check_launch("mask_zero");
    }
}

// Stage 1 of an iteration (SKF_STAGE_CONTRACT): everything that depends on G only.
static void stage_contract(skf_plan* p, hipStream_t st) {
    const bool dfmc = (p->variant == SKF_DFMC);
    if (dfmc && p->first_iter) zero_unknown_entries(p, st);
    p->first_iter = false;

    std::vector<int> all;
    // ---- phase A (second stream when available): Gram_i and K_i = pinv(Gram_i).  They depend
    // only on G, exactly like the relation contractions of phase B, so the two phases overlap.
    // The Gram products use the whole chip and stay on the main stream; the pseudo-inverses are
    // three single-workgroup kernels with a long serial chain: they go to the second stream and
    // run underneath the relation contractions.
    // (Round 1 / 2 A-B: with the Gram products on the second stream as well the contractions slow down by what those
    // kernels take -- profiles/r02_pipeline_ab.txt.)
    hipStream_t sa = st;
    for (size_t i = 0; i < p->types.size(); ++i) {
        gram(p, p->types[i], 1, st);
        all.push_back((int)i);
    }
    if (p->overlap) {
        SKF_HIP(hipEventRecord(p->ev_fork, st));
        SKF_HIP(hipStreamWaitEvent(p->aux, p->ev_fork, 0));
        sa = p->aux;
    }
    plan_pinv(p, all, sa);
    if (p->sw.debug_pinv) {
        // diagnostics: verdict of the Cholesky fast path and the diagonal range of every Gram matrix
        SKF_HIP(hipStreamSynchronize(sa));
        std::vector<int> ok(p->types.size());
        SKF_HIP(hipMemcpy(ok.data(), p->eigOk.ptr, ok.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < p->types.size(); ++i) {
            const TypeState& t = p->types[i];
            std::vector<double> gm((size_t)t.c * t.c);
            SKF_HIP(hipMemcpy(gm.data(), t.Gram.ptr, gm.size() * 8, hipMemcpyDeviceToHost));
            double lo = 1e300, hi = 0.0;
            for (int k = 0; k < t.c; ++k) {
                lo = gm[(size_t)k * t.c + k] < lo ? gm[(size_t)k * t.c + k] : lo;
                hi = gm[(size_t)k * t.c + k] > hi ? gm[(size_t)k * t.c + k] : hi;
            }
            fprintf(stderr, "[skf pinv] type %zu c=%d chol_ok=%d diag min %.3e max %.3e\n", i, t.c, ok[i], lo, hi);
        }
    }
    if (p->overlap) SKF_HIP(hipEventRecord(p->ev_join, p->aux));

    // ---- phase B (main stream): every product that streams a relation matrix, and W = G_i^T P
    for (RelState& r : p->rels) {
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        // A masked DFMC relation is contracted twice per iteration: here, before its completion, only
        // W = G_i^T R G_j is needed (_dfmc.py:311-314), and the narrower factor does it -- with c_i < c_j as
        // W = (R^T G_i)^T G_j (config 5, user x movie: rank 128 instead of 256 through the 8 GB relation).
        const bool w_by_q = dfmc && r.masked && ti.c < tj.c;
        if (r.kn) {                 // known entries only: W from the stored residuals and c x c cross-Gram products
            known_w(p, r, st);
            continue;
        }
        if (r.absent) {
            SKF_HIP(hipMemsetAsync(r.W.ptr, 0, r.W.bytes, st));
        } else if (w_by_q) {
            contraction_Q(p, r, st);
            GemmArgs g = gemm_args(r.Q.ptr, 1, ti.c, tj.G.ptr, tj.c, 1, r.W.ptr, tj.c, ti.c, tj.c, (int)tj.n, EPI_STORE, 0);
            wide_gemm(p, g, st);
        } else {
            contraction_P(p, r, st);
        }
        if (!(dfmc && r.masked)) contraction_Q(p, r, st);     // a masked relation is completed first
        if (!r.absent && !w_by_q) {
            GemmArgs g = gemm_args(rows_of(p, ti.G, ti, r.r0), 1, ti.c, r.P.ptr, tj.c, 1, r.W.ptr, tj.c, ti.c, tj.c,
                                   (int)r.nr, EPI_STORE, 0);
            wide_gemm(p, g, st);
        }
    }
    if (p->overlap) SKF_HIP(hipStreamWaitEvent(st, p->ev_join, 0));
}

// SKF_BF16: one elementwise pass over the stored relation against its reconstruction H G_j^T (r.H = G_i S must
// be current): DFMC completion of the unknown entries, or the squared residual into p->sqpart (one f64 per tile)
enum { MODE_COMPLETE = 0, MODE_SQERR = 1 };
static void tile_epilogue_operands(skf_plan* p, RelState& r, hipStream_t st) {      // bf16 H and G_j of the tile kernels
    TypeState& tj = p->types[r.col];
    const int nr = (int)r.nr, nj = (int)tj.n, cj = tj.c;
    launch_to_bf16<float>((uint16_t*)r.Hb.ptr, r.ldhb, (const float*)r.H.ptr, (int64_t)cj, nr, cj, false, st);
    launch_to_bf16<float>((uint16_t*)r.Gb.ptr, r.ldhb, (const float*)tj.G.ptr, (int64_t)cj, nj, cj, false, st);
}

static void launch_tile_epilogue(skf_plan* p, RelState& r, int mode, hipStream_t st, bool operands = true) {
    TypeState& tj = p->types[r.col];
    const int nr = (int)r.nr, nj = (int)tj.n;
    if (operands) tile_epilogue_operands(p, r, st);
    Bf16GemmArgs g;
    memset(&g, 0, sizeof g);
    // transposed product: tile rows = relation columns (A = bf16 G_j), tile columns = relation rows (Bt = bf16 H)
    g.A = (const uint16_t*)r.Gb.ptr; g.Bt = (const uint16_t*)r.Hb.ptr;
    g.lda = r.ldhb; g.ldb = r.ldhb;
    g.M = nj; g.N = nr; g.Kp = (int)r.ldhb; g.k_chunk = (int)r.ldhb;
    g.a_kstep = 64; g.b_kstep = 64;
    g.R = (uint16_t*)r.Rb.ptr; g.ldr = r.ldrb;
    if (r.binary) { g.Rbits = (const uint8_t*)r.Bb.ptr; g.ldrbits = r.ldbb; }
    g.mbits = (const uint8_t*)r.Mb.ptr; g.ldmb = r.ldmb;
    g.sq = (double*)p->sqpart.ptr;
    if (mode == MODE_COMPLETE && r.use_klist) {
        g.koff = (const uint32_t*)r.Koff.ptr;
        g.klist = (const uint32_t*)r.Klist.ptr;
    }
    if (mode == MODE_COMPLETE) {
        // 128 relation columns x 256 relation rows per workgroup, 256 threads, 68 KiB of LDS: two workgroups per CU, one
        // tile's write-out under the other's K loop (3.9 vs 4.7 ms at config 5).  The residual pass only reads the
        // relation and is faster on the 256 x 256 tile (4.7 vs 4.9 ms): it stays there.
        dim3 grid(cdiv(nr, 256), cdiv(nj, 128));
        const int smem = 256 * (128 + 8) * 2;
        static DeviceOnce once;
        allow_dynamic_lds(once, gemm_bf16_kernel<256, 0, EPI_T_COMPLETE>, smem);
        hipLaunchKernelGGL((gemm_bf16_kernel<256, 0, EPI_T_COMPLETE>), grid, dim3(256), smem, st, g);
        check_launch("tile_epilogue_bf16");
        return;
    }
    dim3 grid(cdiv(nr, 256), cdiv(nj, 256));
    const int smem = (3 * 256 + 2 * 256) * 8 * 16;
    static DeviceOnce once;
    allow_dynamic_lds(once, gemm_bf16_v2_kernel<256, 0, false, EPI_T_SQERR>, smem);
    hipLaunchKernelGGL((gemm_bf16_v2_kernel<256, 0, false, EPI_T_SQERR>), grid, dim3(512), smem, st, g);
    check_launch("tile_epilogue_bf16");
}

// Stage 2 (SKF_STAGE_BACKBONE): S = K_i W K_j; DFMC: completion, then P and Q of masked relations.
// every rank <= SMALLC: the c x c chains run in the one-workgroup kernels (SKF_NO_SMALL_CHAIN=1: off)
static bool small_chain(const skf_plan* p) {
    if (p->sw.no_small_chain) return false;
    for (const TypeState& t : p->types)
        if (t.c > SMALLC) return false;
    return true;
}

static void stage_backbone(skf_plan* p, hipStream_t st) {
    const bool dfmc = (p->variant == SKF_DFMC);
    const bool chain = small_chain(p);
    if (chain) {
        for (size_t k0 = 0; k0 < p->rels.size(); k0 += CHAIN_MAXB) {
            BackboneBatch bb;
            int nb = 0;
            for (size_t k = k0; k < p->rels.size() && nb < CHAIN_MAXB; ++k, ++nb) {
                RelState& r = p->rels[k];
                bb.Ki[nb] = (const double*)p->types[r.row].K.ptr;
                bb.Kj[nb] = (const double*)p->types[r.col].K.ptr;
                bb.W[nb] = (const double*)r.W.ptr;
                bb.S[nb] = (double*)r.S.ptr;
                bb.ci[nb] = p->types[r.row].c;
                bb.cj[nb] = p->types[r.col].c;
            }
            size_t smem = 0;
            for (int q = 0; q < nb; ++q) {
                const size_t need = ((size_t)bb.ci[q] * bb.ci[q] + 2 * (size_t)bb.ci[q] * bb.cj[q] + (size_t)bb.cj[q] * bb.cj[q]) * 8;
                if (need > smem) smem = need;
            }
            static DeviceOnce once_bb, once_bt;
            allow_dynamic_lds(once_bb, backbone_small_kernel, 4 * SMALLC * SMALLC * 8);
            allow_dynamic_lds(once_bt, bterms_small_kernel, 4 * SMALLC * SMALLC * 8);
            hipLaunchKernelGGL(backbone_small_kernel, dim3(nb), dim3(256), smem, st, bb);
            check_launch("backbone_small");
        }
    }
    for (RelState& r : p->rels) {
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int nr = (int)r.nr, nj = (int)tj.n, ci = ti.c, cj = tj.c;
        GemmArgs g;
        if (!chain) {
            // T1 = K_i W ; S = T1 K_j
            g = gemm_args(ti.K.ptr, ci, 1, r.W.ptr, cj, 1, r.T1.ptr, cj, ci, cj, ci, EPI_STORE, 0);
            small_gemm(p, g, st);
            g = gemm_args(r.T1.ptr, cj, 1, tj.K.ptr, cj, 1, r.S.ptr, cj, ci, cj, cj, EPI_STORE, 1);
            small_gemm(p, g, st);
        }
        if (!(dfmc && r.masked)) continue;
        if (r.kn) {                 // completion, P and Q of _dfmc.py:319-325, 341-345 on the known entries
            known_operands(p, r, st, false);
            known_row_pass(p, r, st);
            known_col_pass(p, r, st);
            continue;
        }
        if (!r.absent) {
            // H = G_i[blk] S ; Rw[mask] = (H G_j^T)[mask] ; P = Rw G_j        (_dfmc.py:319-325)
            g = gemm_args(rows_of(p, ti.G, ti, r.r0), ci, 1, r.S.ptr, cj, 1, r.H.ptr, cj, nr, cj, ci, EPI_STORE, 0);
            mixed_gemm(p, g, st);
            if (r.mask) {
                if (p->bf16) {
                    // completed entries go to the one stored copy as bf16: bf16 operands on the matrix cores,
                    // the write-out is the bound (gemm_bf16_v2_kernel<.., EPI_T_COMPLETE>)
                    launch_tile_epilogue(p, r, MODE_COMPLETE, st);
                } else {
                    g = gemm_args(r.H.ptr, cj, 1, tj.G.ptr, 1, cj, r.Rw.ptr, r.ldr, nr, nj, cj, EPI_MASKED_STORE, 0);
                    g.mask = (const uint8_t*)r.Mb.ptr;
                    g.ldmask = r.ldmb;
                    g.mask_bits = 1;
                    plan_gemm(p, g, st);
                }
            }
            contraction_P(p, r, st);
        }
        contraction_Q(p, r, st);
    }
}

// Stage 3 (SKF_STAGE_ACCUMULATE): the E / D sums of this plan's row blocks, column sides and constraints.
// The G_i B^-+ terms of _dfmf.py:278-282 are linear in B: they are added once per type with the
// sums of B^+ / B^- over the relations (half the n x c x c work of adding them relation by relation).
static void stage_accumulate(skf_plan* p, hipStream_t st) {
    const bool dfmc = (p->variant == SKF_DFMC);
    const int nan_upd = dfmc ? 0 : 1;       // _update_G_for_Rij (_dfmc.py:127-178) has no nan_to_num
    const bool fused = (p->engine == SKF_ENGINE_MFMA);
    const size_t nt = p->types.size();
    std::vector<char> touched(nt, 0);       // E/D of the type already written this iteration
    std::vector<int> sides_left(nt, 0);     // E/D side products still to come for the type
    for (RelState& r : p->rels) {
        if (!r.absent) sides_left[r.row] += 1;
        if (r.col_side) sides_left[r.col] += 1;
    }
    for (size_t i = 0; i < nt; ++i) {
        TypeState& t = p->types[i];
        // the fused update overwrites E/D on first touch -- of whole matrices only
        if (!fused || sides_left[i] == 0 || p->sliced) {
            SKF_HIP(hipMemsetAsync(t.E.ptr, 0, t.E.bytes, st));
            SKF_HIP(hipMemsetAsync(t.D.ptr, 0, t.D.bytes, st));
            touched[i] = 1;
        }
    }
    SKF_HIP(hipMemsetAsync((char*)p->ws_base + p->btot_off, 0, p->btot_bytes, st));
    // sum_r B_r^+- per type.  A plan with row blocks lists every relation and owns a share of the rows
    // of every type: it sums over all relations; otherwise over the plan's own relations.
    const bool chain = small_chain(p);
    for (RelState& r : p->rels) {
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        if (chain) {
            BTermsArgs ba;
            ba.S = (const double*)r.S.ptr; ba.Gram_i = (const double*)ti.Gram.ptr; ba.Gram_j = (const double*)tj.Gram.ptr;
            ba.Bp_i = (double*)ti.Bp_tot.ptr; ba.Bn_i = (double*)ti.Bn_tot.ptr;
            ba.Bp_j = (double*)tj.Bp_tot.ptr; ba.Bn_j = (double*)tj.Bn_tot.ptr;
            ba.ci = ti.c; ba.cj = tj.c; ba.nan_to_num = nan_upd;
            const size_t smem = ((size_t)ti.c * ti.c + 2 * (size_t)ti.c * tj.c + (size_t)tj.c * tj.c) * 8;
            hipLaunchKernelGGL(bterms_small_kernel, dim3(1), dim3(256), smem, st, ba);
            check_launch("bterms_small");
            continue;
        }
        relation_small_terms(p, r, nan_upd, EPI_SPLIT_ACC, ti.Bp_tot.ptr, ti.Bn_tot.ptr, tj.Bp_tot.ptr, tj.Bn_tot.ptr,
                             true, true, st);
    }
    // c x c operands of the fused side products in the master type: the f32 engines round S of every
    // relation with a side here and sum B+- of every type in ONE batched launch
    std::vector<const void*> Bn_m(nt), Bp_m(nt), S_m(p->rels.size());
    {
        const bool cast = !p->f64 && fused;
        CastBatch cb;
        int ne = 0, max_count = 1;
        bool overflow = false;
        auto add = [&](const Slot& src, const Slot& dst, int count) -> const void* {
            if (!cast) return src.ptr;
            if (ne >= CAST_MAXB) {                   // very large graphs: one launch per matrix
                overflow = true;
                hipLaunchKernelGGL((cast_kernel<float, double>), dim3(elem_grid(count)), dim3(256), 0, st, (float*)dst.ptr,
                                   (int64_t)count, (const double*)src.ptr, (int64_t)count, (int64_t)1, (int64_t)count);
                check_launch("cast");
                return dst.ptr;
            }
            cb.src[ne] = (const double*)src.ptr;
            cb.dst[ne] = (float*)dst.ptr;
            cb.count[ne] = count;
            if (count > max_count) max_count = count;
            ++ne;
            return dst.ptr;
        };
        for (size_t i = 0; i < nt; ++i) {
            TypeState& t = p->types[i];
            Bn_m[i] = add(t.Bn_tot, t.Bn32, t.c * t.c);
            Bp_m[i] = add(t.Bp_tot, t.Bp32, t.c * t.c);
        }
        for (size_t k = 0; k < p->rels.size(); ++k) {
            RelState& r = p->rels[k];
            S_m[k] = (!r.absent || r.col_side) ? add(r.S, r.S32, p->types[r.row].c * p->types[r.col].c) : r.S.ptr;
        }
        (void)overflow;
        if (cast && ne > 0) {
            hipLaunchKernelGGL(cast_batched_kernel, dim3(elem_grid(max_count), ne), dim3(256), 0, st, cb);
            check_launch("cast_batched");
        }
    }
    // type-level term on rows [t0, t0 + tn): separately (row blocks / VALU engine), or inside the
    // last side product of the type (fused engine on whole matrices)
    auto type_term = [&](size_t i) {
        TypeState& t = p->types[i];
        if (t.tn <= 0) return;
        void* G = rows_of(p, t.G, t, t.t0);
        void* E = rows_of(p, t.E, t, t.t0);
        void* D = rows_of(p, t.D, t, t.t0);
        if (fused) {
            side_update(p, nullptr, 0, 0, nullptr, 0, 0, t, G, E, D, (int)t.tn, Bn_m[i], Bp_m[i], true,
                        touched[i] != 0, 0, st);
        } else {
            GemmArgs g = gemm_args(G, t.c, 1, t.Bn_tot.ptr, t.c, 1, E, t.c, (int)t.tn, t.c, t.c, EPI_ACC, 0);
            mixed_gemm(p, g, st);
            g = gemm_args(G, t.c, 1, t.Bp_tot.ptr, t.c, 1, D, t.c, (int)t.tn, t.c, t.c, EPI_ACC, 0);
            mixed_gemm(p, g, st);
        }
        touched[i] = 1;
    };
    const bool fuse_type_term = fused && !p->sliced;
    for (size_t rk = 0; rk < p->rels.size(); ++rk) {
        RelState& r = p->rels[rk];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int nr = (int)r.nr, nj = (int)tj.n, ci = ti.c, cj = tj.c;
        const bool row_side = !r.absent, col_side = r.col_side;
        void* Gi = rows_of(p, ti.G, ti, r.r0);
        void* Ei = rows_of(p, ti.E, ti, r.r0);
        void* Di = rows_of(p, ti.D, ti, r.r0);
        GemmArgs g;
        if (fused) {
            // row side: A = P S^T (Sop(k,j) = S[j][k]);  column side: C = Q S
            if (!row_side && !col_side) continue;
            const void* Sm = S_m[rk];
            if (row_side && r.kn) {
                const bool last = fuse_type_term && --sides_left[r.row] == 0;
                known_row_side(p, r, touched[r.row] != 0, st, false);
                touched[r.row] = 1;
                if (last) type_term(r.row);
            } else if (row_side) {
                const bool last = fuse_type_term && --sides_left[r.row] == 0;
                side_update(p, r.P.ptr, cj, cj, Sm, 1, cj, ti, Gi, Ei, Di, nr, Bn_m[r.row], Bp_m[r.row], last,
                            touched[r.row] != 0, nan_upd, st);
                touched[r.row] = 1;
            }
            if (col_side) {
                const bool last = fuse_type_term && --sides_left[r.col] == 0;
                side_update(p, r.Q.ptr, ci, ci, Sm, cj, 1, tj, tj.G.ptr, tj.E.ptr, tj.D.ptr, nj, Bn_m[r.col],
                            Bp_m[r.col], last, touched[r.col] != 0, nan_upd, st);
                touched[r.col] = 1;
            }
            continue;
        }
        if (row_side && r.kn) {
            known_row_side(p, r, true, st, false);
        } else if (row_side) {
            // E_i += (P S^T)+ ; D_i += (P S^T)-          (_dfmf.py:254-258, 278-279)
            g = gemm_args(r.P.ptr, cj, 1, r.S.ptr, 1, cj, Ei, ci, nr, ci, cj, EPI_SPLIT_ACC, nan_upd);
            g.C2 = Di;
            mixed_gemm(p, g, st);
        }
        if (col_side) {
            // E_j += (Q S)+ ; D_j += (Q S)-              (_dfmf.py:266-270, 281-282)
            g = gemm_args(r.Q.ptr, ci, 1, r.S.ptr, cj, 1, tj.E.ptr, cj, nj, cj, ci, EPI_SPLIT_ACC, nan_upd);
            g.C2 = tj.D.ptr;
            mixed_gemm(p, g, st);
        }
    }
    if (!fuse_type_term)
        for (size_t i = 0; i < nt; ++i) type_term(i);     // E_i += G_i sum B- ; D_i += G_i sum B+
    theta_terms(p, st);
}

// E/D accumulation of one iteration: everything of the loop body except the final G update.
// With relation sharding every rank runs this on ITS relations / constraints, the E and D
// accumulators are then summed over the ranks (one all-reduce), and apply_update follows.
static void accumulate_fit(skf_plan* p, hipStream_t st) {
    stage_contract(p, st);
    stage_backbone(p, st);
    stage_accumulate(p, st);
}

// G_i <- G_i * sqrt(E_i / max(D_i, eps)) of one type   (_dfmf.py:294-296)
static void update_type(skf_plan* p, TypeState& t, hipStream_t st) {
    if (p->bf16 && t.n > 0) {                  // update and G^T refresh in one pass
        hipLaunchKernelGGL(mult_update_transpose_kernel, dim3((unsigned)cdiv(t.c, 32), (unsigned)cdiv(t.n, 32)), dim3(256), 0,
                           st, (float*)t.G.ptr, (const float*)t.E.ptr, (const float*)t.D.ptr, (int64_t)t.n, (int64_t)t.c,
                           (uint16_t*)t.GTb.ptr, t.ldgt, (uint16_t*)t.Grow.ptr, t.ldrow);
        check_launch("mult_update_transpose");
        return;
    }
    mult_update(p, t, st);
    refresh_gt(p, t, st);
}

// ... of every type (`done`: types the schedule has updated already)
static void apply_update(skf_plan* p, hipStream_t st, const std::vector<char>* done = nullptr) {
    for (TypeState& t : p->types)          // known-entries relations: the factors their stored residuals belong to
        if (t.keep_prev) SKF_HIP(hipMemcpyAsync(t.Gp.ptr, t.G.ptr, t.G.bytes, hipMemcpyDeviceToDevice, st));
    p->kn_first = false;
    for (size_t i = 0; i < p->types.size(); ++i)
        if (!(done && (*done)[i])) update_type(p, p->types[i], st);
}



// ------------------------------------------------------------------------------------------
// Collectives behind the boundary (include/skfusion_hip.h, skf_comm_*): RCCL bound at run time with dlopen -- no
// torch, no link-time dependency -- or caller-supplied callbacks (CPU tests over gloo, other transports).
// ------------------------------------------------------------------------------------------
struct NcclUniqueId { char internal[128]; };
typedef int (*nccl_get_unique_id_t)(NcclUniqueId*);
typedef int (*nccl_comm_init_rank_t)(void**, int, NcclUniqueId, int);
typedef int (*nccl_comm_destroy_t)(void*);
typedef int (*nccl_all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_reduce_scatter_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_all_gather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*nccl_get_error_string_t)(int);
typedef int (*nccl_comm_count_t)(const void*, int*);
struct Rccl {
    nccl_comm_count_t comm_count = nullptr;          // optional (skf_comm_info)
    nccl_comm_count_t comm_user_rank = nullptr;
    void* handle = nullptr;
    nccl_get_unique_id_t get_unique_id = nullptr;
    nccl_comm_init_rank_t comm_init_rank = nullptr;
    nccl_comm_destroy_t comm_destroy = nullptr;
    nccl_all_reduce_t all_reduce = nullptr;
    nccl_reduce_scatter_t reduce_scatter = nullptr;
    nccl_all_gather_t all_gather = nullptr;
    nccl_get_error_string_t error_string = nullptr;
};
static Rccl g_rccl;
static std::once_flag g_rccl_once;

// librccl of the process if one is loaded already (PyTorch-ROCm bundles its own), else SKF_RCCL_PATH / the loader path / ROCm
static const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        const char* names[] = {env_str("SKF_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void* h = nullptr;
        for (const char* nm : names)
            if (nm && !h) h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* nm : names)
            if (nm && !h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        g_rccl.handle = h;
        g_rccl.get_unique_id = (nccl_get_unique_id_t)dlsym(h, "ncclGetUniqueId");
        g_rccl.comm_init_rank = (nccl_comm_init_rank_t)dlsym(h, "ncclCommInitRank");
        g_rccl.comm_destroy = (nccl_comm_destroy_t)dlsym(h, "ncclCommDestroy");
        g_rccl.all_reduce = (nccl_all_reduce_t)dlsym(h, "ncclAllReduce");
        g_rccl.reduce_scatter = (nccl_reduce_scatter_t)dlsym(h, "ncclReduceScatter");
        g_rccl.all_gather = (nccl_all_gather_t)dlsym(h, "ncclAllGather");
        g_rccl.error_string = (nccl_get_error_string_t)dlsym(h, "ncclGetErrorString");
        g_rccl.comm_count = (nccl_comm_count_t)dlsym(h, "ncclCommCount");
        g_rccl.comm_user_rank = (nccl_comm_count_t)dlsym(h, "ncclCommUserRank");
    });
    if (!g_rccl.handle || !g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce ||
        !g_rccl.reduce_scatter || !g_rccl.all_gather)
        SKF_FAIL(SKF_E_STATE, "librccl.so could not be loaded (set SKF_RCCL_PATH, or use skf_comm_create_callback)");
    return g_rccl;
}

}  // namespace skf

struct skf_comm {
    int rank = 0, world = 1;
    void* nccl = nullptr;                  // ncclComm_t, or
    skf_collective_fn fn = nullptr;        // caller-supplied collectives
    void* user = nullptr;
    bool null_comm = false;                // collectives skipped (skf_comm_create_null)
};

namespace skf {

enum { COLL_ALL_REDUCE = 0, COLL_REDUCE_SCATTER = 1, COLL_ALL_GATHER = 2 };

// op over `count` elements per rank (COLL_ALL_REDUCE: the whole buffer; the other two: chunk `rank` of world * count), in place
static void collective(skf_comm* c, int op, void* buf, size_t count, int dtype, hipStream_t st) {
    if (!c) SKF_FAIL(SKF_E_STATE, "no communicator attached to the plan (skf_plan_set_comm)");
    if (count == 0) return;
    if (c->fn) {
        const int rc = c->fn(c->user, op, buf, count, dtype, (void*)st);
        if (rc != 0) SKF_FAIL(SKF_E_HIP, "collective callback failed (op %d, status %d)", op, rc);
        return;
    }
    if (c->null_comm) {
        // timing runs of ONE rank of a sharded fit (skf_comm_create_null): nothing is exchanged; a sum over the ranks is stood
        // in for by `world` times this rank's partial, so that the Gram matrices, backbones and updates the timed launches see
        // stay in the range of a real run (a Gram matrix of 1/8 of the rows sends the factors off scale, and the
        // pseudo-inverses then take their slow fallbacks); gathered rows of other ranks keep their old values
        if (op == COLL_ALL_GATHER || c->world == 1 || dtype == SKF_BF16) return;
        const size_t es = dtype == SKF_F64 ? 8 : 4;
        char* at = (char*)buf + (op == COLL_REDUCE_SCATTER ? (size_t)c->rank * count * es : 0);
        if (dtype == SKF_F64)
            hipLaunchKernelGGL((scale_kernel<double>), dim3(elem_grid((int64_t)count)), dim3(256), 0, st, (double*)at, (int64_t)count, (double)c->world);
        else
            hipLaunchKernelGGL((scale_kernel<float>), dim3(elem_grid((int64_t)count)), dim3(256), 0, st, (float*)at, (int64_t)count, (float)c->world);
        check_launch("scale (null communicator)");
        return;
    }
    if (c->world == 1 && !c->nccl) return;          // a single rank without a transport: nothing to exchange
    const Rccl& r = rccl();
    const int nt = dtype == SKF_F64 ? 8 /* ncclFloat64 */ : 7 /* ncclFloat32 */;
    const size_t es = dtype == SKF_F64 ? 8 : dtype == SKF_BF16 ? 2 : 4;
    char* mine = (char*)buf + (size_t)c->rank * count * es;
    int rc = 0;
    if (dtype == SKF_BF16 && op != COLL_ALL_GATHER) SKF_FAIL(SKF_E_INVALID, "bf16 collectives: all-gather only");
    if (op == COLL_ALL_REDUCE) rc = r.all_reduce(buf, buf, count, nt, 0 /* ncclSum */, c->nccl, st);
    else if (op == COLL_REDUCE_SCATTER) rc = r.reduce_scatter(buf, mine, count, nt, 0, c->nccl, st);
    else if (dtype == SKF_BF16) rc = r.all_gather(mine, buf, count * 2, 1 /* ncclUint8: the bytes of the bf16 rows */, c->nccl, st);
    else rc = r.all_gather(mine, buf, count, nt, c->nccl, st);
    if (rc != 0) SKF_FAIL(SKF_E_HIP, "RCCL collective %d failed: %s", op, r.error_string ? r.error_string(rc) : "?");
}

// elements per rank when `total` elements are cut into `world` equal ranges (multiples of 64; the region's pad takes the rest)
static size_t flat_chunk(size_t total, int world) { return ((total + world - 1) / world + 63) / 64 * 64; }

// One iteration with the exchanges inside: plans without row blocks run accumulate -> exchange -> update, plans with row
// blocks the four stages.  E / D: reduce-scatter (every rank ends with the sums of ITS element range), update of that
// range of G, all-gather of G -- (2 + 1) * (world - 1) / world of the factor bytes per rank instead of the
// 2 * 2 * (world - 1) / world of an all-reduce of both accumulators.
static void exchange_and_update(skf_plan* p, hipStream_t st) {
    skf_comm* c = p->comm;
    const size_t total = p->flat_bytes / p->esz;
    const size_t chunk = flat_chunk(total, c->world);
    if (chunk * c->world * p->esz > p->flat_bytes + 64 * 1024) SKF_FAIL(SKF_E_STATE, "exchange ranges exceed the region pad (world %d)", c->world);
    char* base = (char*)p->ws_base;
    void* E = base + p->flat_e_off;
    void* D = base + p->flat_d_off;
    void* G = base + p->flat_g_off;
    collective(c, COLL_REDUCE_SCATTER, E, chunk, p->mt, st);
    collective(c, COLL_REDUCE_SCATTER, D, chunk, p->mt, st);
    for (TypeState& t : p->types)          // known-entries relations: the factors their stored residuals belong to
        if (t.keep_prev) SKF_HIP(hipMemcpyAsync(t.Gp.ptr, t.G.ptr, t.G.bytes, hipMemcpyDeviceToDevice, st));
    p->kn_first = false;
    const size_t lo = (size_t)c->rank * chunk;
    if (p->f64)
        hipLaunchKernelGGL((mult_update_kernel<double>), dim3(elem_grid((int64_t)chunk)), dim3(256), 0, st, (double*)G + lo,
                           (const double*)E + lo, (const double*)D + lo, (int64_t)chunk, 1, (int64_t)1, (int64_t)1);
    else
        hipLaunchKernelGGL((mult_update_kernel<float>), dim3(elem_grid((int64_t)chunk)), dim3(256), 0, st, (float*)G + lo,
                           (const float*)E + lo, (const float*)D + lo, (int64_t)chunk, 1, (int64_t)1, (int64_t)1);
    check_launch("mult_update(range)");
    collective(c, COLL_ALL_GATHER, G, chunk, p->mt, st);
    for (TypeState& t : p->types) refresh_gt(p, t, st);
}

static void iterate_dist(skf_plan* p, hipStream_t st) {
    skf_comm* c = p->comm;
    char* base = (char*)p->ws_base;
    if (!p->sliced) {
        accumulate_fit(p, st);
    } else {
        stage_contract(p, st);
        collective(c, COLL_ALL_REDUCE, base + p->xw_off, p->xw_bytes / 8, SKF_F64, st);
        collective(c, COLL_ALL_REDUCE, base + p->xq_off, p->xq_bytes / p->esz, p->mt, st);
        stage_backbone(p, st);
        collective(c, COLL_ALL_REDUCE, base + p->xqm_off, p->xqm_bytes / p->esz, p->mt, st);
        stage_accumulate(p, st);
    }
    exchange_and_update(p, st);
}

// bytes one rank SENDS per iteration of skf_iterate_dist on a ring (reduce-scatter / all-gather: (world-1)/world of the
// buffer, all-reduce: twice that)
static size_t exchange_bytes(const skf_plan* p, int world) {
    if (world <= 1) return 0;
    const double f = (double)(world - 1) / world;
    if (p->owned) {
        // reduce-scatter of every relation's partial Q, all-gather of every type's updated rows (master type, or the bf16
        // operand rows), all-reduce of the c x c Gram and W partials
        double b = 0.0;
        for (const RelState& r : p->rels) b += f * (double)p->types[r.col].n_alloc * p->types[r.row].c * (double)p->esz;
        for (const TypeState& t : p->types)
            b += f * (double)t.n_alloc * (t.gather_master ? (double)t.c * (double)p->esz : (double)t.ldrow * 2.0);
        b += 2.0 * f * (double)p->xg_bytes;                 // (the Gram range goes out as a whole, slot padding included)
        for (const RelState& r : p->rels) b += 2.0 * f * 8.0 * (double)p->types[r.row].c * p->types[r.col].c;
        return (size_t)(b + 0.5);
    }
    double b = 3.0 * f * (double)(flat_chunk(p->flat_bytes / p->esz, world) * world * p->esz);
    if (p->sliced) b += 2.0 * f * (double)(p->xw_bytes + p->xq_bytes + p->xqm_bytes);
    return (size_t)b;
}

// ------------------------------------------------------------------------------------------
// SKF_OPT_OWNED_ROWS: one iteration of a fit whose row blocks follow the OWNERSHIP of the factor rows (every rank holds the
// rows [t0, t0 + tn) of every type: of its factor, of E / D, of every relation with that row type).  Replaces the
// reference's per-block tasks (_dfmf.py:69-73, _dfmc.py:341-345).  What crosses ranks:
//     Gram_t = sum over ranks of G_t[own]^T G_t[own]            all-reduce, c x c f64, all types at once
//     W_r    = sum over ranks of G_i[own]^T (R_blk G_j)         all-reduce, c_i x c_j f64, per relation
//     Q_r    = sum over ranks of R_blk^T G_i[own]               REDUCE-SCATTER to the owners of type j (raw: the +- split of
//                                                               _dfmf.py:268-270 is not linear), per relation
//     G_t                                                       ALL-GATHER of the updated owned rows, per type (SKF_BF16: of
//                                                               their bf16 operand copy unless a constraint reads the f32 rows)
// Row sides (P S^T)+-, type terms G_i sum B-+, constraint rows and the update act on owned rows: no exchange of E / D.
// Three streams when the graph allows (ranks in 65 .. 512, sparse constraints only -- the conditions under which the c x c
// chains need no split-K scratch): main = Gram, contractions, W; second = pseudo-inverses, backbones, side products,
// updates; `cs` = the exchanges.  A relation's exchanges run under the next relation's contractions, a type is updated and
// gathered as soon as its last relation's sides are in.  Otherwise everything is issued in the same order on one stream.
// ------------------------------------------------------------------------------------------
static bool owned_can_overlap(const skf_plan* p) {
    if (!p->overlap || p->engine != SKF_ENGINE_MFMA) return false;
    for (const ThetaState& th : p->thetas)
        if (!th.sparse) return false;
    // ranks above 512 (DFMC: 256): a c x c product on the second stream could ask for the main stream's split-K scratch;
    // every rank <= 64: launch-bound graphs, one stream.  Mixed ranks (config 5: 16 ... 256) do overlap: the replicated
    // pseudo-inverses (0.65 ms at order 256) then run beside the list passes instead of in front of them.
    const int cmax = (p->variant == SKF_DFMC) ? 256 : 512;
    bool all_small = true;
    for (const TypeState& t : p->types) {
        if (t.c > cmax) return false;
        if (t.c > SMALLC) all_small = false;
    }
    return !all_small;
}

static void iterate_owned(skf_plan* p, hipStream_t st) {
    skf_comm* c = p->comm;
    const size_t nt = p->types.size(), nr = p->rels.size();
    const bool dfmc = (p->variant == SKF_DFMC);
    const int nan_upd = dfmc ? 0 : 1;
    const bool fused = (p->engine == SKF_ENGINE_MFMA);
    const bool multi = owned_can_overlap(p);
    hipStream_t ax = multi ? p->aux : st;
    hipStream_t cs = (multi && p->cs) ? p->cs : st;
    char* base = (char*)p->ws_base;
    if (dfmc && p->first_iter) zero_unknown_entries(p, st);
    p->first_iter = false;
    // events: [0] Gram partials, [1] Gram sums; per relation: W partial, W sum, Q partial, Q scattered, backbone (+ completion
    // operands), P after the completion; per type: rows updated, rows gathered, operand copies refreshed
    const size_t n_ev = 2 + 6 * nr + 3 * nt;
    while (p->ev_own.size() < n_ev) {
        hipEvent_t e;
        SKF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->ev_own.push_back(e);
    }
    enum { R_W = 0, R_WX = 1, R_Q = 2, R_QX = 3, R_S = 4, R_P2 = 5, T_UPD = 0, T_G = 1, T_GT = 2 };
    auto ev_r = [&](size_t k, int what) { return p->ev_own[2 + 6 * k + what]; };
    auto ev_t = [&](size_t i, int what) { return p->ev_own[2 + 6 * nr + 3 * i + what]; };
    auto rec = [&](hipEvent_t e, hipStream_t s) { SKF_HIP(hipEventRecord(e, s)); };
    auto wait = [&](hipStream_t s, hipEvent_t e) { SKF_HIP(hipStreamWaitEvent(s, e, 0)); };
    auto own = [&](const Slot& s, const TypeState& t) { return rows_of(p, s, t, t.t0); };

    // ---- Gram partials over the owned rows (main stream, at the head: the chain Gram -> sum -> pseudo-inverses (~1.2 ms at
    // rank 256, one workgroup per type) -> backbones -> side products IS the critical path of a rank, the contractions run
    // beside it), their sum (exchange stream), the pseudo-inverses (second stream).
    // (Measured, rank 3 of 8 at config 3: with the partials on the second stream underneath the first contraction their
    // split-K reduce crawled for 0.26 ms, the pseudo-inverses started 0.33 ms later and the iteration took as long as
    // before -- 2.55 ms; profiles/r04_owned_rank_timeline.txt.)
    std::vector<int> all;
    hipStream_t sg = st;
    const bool gram_aux = false;
    if (gram_aux) {
        rec(p->ev_fork, st);                       // (factors set / updated on the caller's stream before this call)
        wait(ax, p->ev_fork);
    }
    for (size_t i = 0; i < nt; ++i) {
        TypeState& t = p->types[i];
        all.push_back((int)i);
        if (t.tn <= 0) {
            SKF_HIP(hipMemsetAsync(t.Gram.ptr, 0, t.Gram.bytes, sg));
            continue;
        }
        GemmArgs g = gemm_args(own(t.G, t), 1, t.c, own(t.G, t), t.c, 1, t.Gram.ptr, t.c, t.c, t.c, (int)t.tn, EPI_STORE, 1);
        g.sym = p->sw.gram_sym;
        run_gemm(GemmTypes{SKF_F64, p->mt, p->mt}, p->engine, g, 0, gram_aux ? p->part_aux.ptr : p->part.ptr,
                 gram_aux ? p->part_aux_bytes : p->part_bytes, sg);
    }
    rec(p->ev_own[0], sg);
    wait(cs, p->ev_own[0]);
    collective(c, COLL_ALL_REDUCE, base + p->xg_off, p->xg_bytes / 8, SKF_F64, cs);
    rec(p->ev_own[1], cs);
    wait(ax, p->ev_own[1]);
    plan_pinv(p, all, ax);
    SKF_HIP(hipMemsetAsync(base + p->btot_off, 0, p->btot_bytes, ax));

    std::vector<char> touched(nt, 0);          // E / D of the type's owned rows already written this iteration
    std::vector<int> sides_left(nt, 0);        // side products of the type still to come
    for (const RelState& r : p->rels) {
        sides_left[r.row] += 1;
        sides_left[r.col] += 1;
    }
    std::vector<const void*> Sm(nr, nullptr);  // the backbone in the master type
    std::vector<char> finished(nt, 0);
    const bool chain = small_chain(p);

    // the type is complete: type term, constraint rows, update of the owned rows (second stream), gather (exchange stream)
    auto finish_type = [&](size_t i) {
        TypeState& t = p->types[i];
        finished[i] = 1;
        // known-entries relations: the factor their stored residuals belong to (all rows: every one is current here)
        if (t.keep_prev) SKF_HIP(hipMemcpyAsync(t.Gp.ptr, t.G.ptr, (size_t)t.n * t.c * p->esz, hipMemcpyDeviceToDevice, ax));
        if (t.tn > 0) {
            void* G = own(t.G, t);
            void* E = own(t.E, t);
            void* D = own(t.D, t);
            if (!touched[i]) {
                SKF_HIP(hipMemsetAsync(E, 0, (size_t)t.tn * t.c * p->esz, ax));
                SKF_HIP(hipMemsetAsync(D, 0, (size_t)t.tn * t.c * p->esz, ax));
                touched[i] = 1;
            }
            if (fused) {
                const void* Bn = t.Bn_tot.ptr;
                const void* Bp = t.Bp_tot.ptr;
                if (!p->f64) {
                    cast_b_sums(t, ax);
                    Bn = t.Bn32.ptr;
                    Bp = t.Bp32.ptr;
                }
                side_update(p, nullptr, 0, 0, nullptr, 0, 0, t, G, E, D, (int)t.tn, Bn, Bp, true, true, 0, ax);
            } else {
                GemmArgs g = gemm_args(G, t.c, 1, t.Bn_tot.ptr, t.c, 1, E, t.c, (int)t.tn, t.c, t.c, EPI_ACC, 0);
                mixed_gemm(p, g, ax);
                g = gemm_args(G, t.c, 1, t.Bp_tot.ptr, t.c, 1, D, t.c, (int)t.tn, t.c, t.c, EPI_ACC, 0);
                mixed_gemm(p, g, ax);
            }
            theta_terms_rows(p, (int)i, true, ax);
            // G[own] <- G[own] * sqrt(E / max(D, eps))   (_dfmf.py:294-296); SKF_BF16: with the bf16 copies of those rows
            if (c->null_comm) {
                // rank-emulation runs (skf_comm_create_null): the factors stay as they were set -- with nothing exchanged the
                // updated rows drift away from the rows of the ranks that do not exist, the Gram matrices go singular and the
                // timed pseudo-inverses take their slow fallbacks (measured: 4-15 ms per iteration); the update launch
                // itself (~10 us per type) is what the measurement then leaves out
            } else if (p->bf16) {
                hipLaunchKernelGGL(mult_update_transpose_kernel, dim3((unsigned)cdiv(t.c, 32), (unsigned)cdiv(t.tn, 32)), dim3(256), 0,
                                   ax, (float*)G, (const float*)E, (const float*)D, (int64_t)t.tn, (int64_t)t.c,
                                   (uint16_t*)t.GTb.ptr + t.t0, t.ldgt, (uint16_t*)t.Grow.ptr + t.t0 * t.ldrow, t.ldrow);
                check_launch("mult_update_transpose(rows)");
            } else if (p->f64) {
                hipLaunchKernelGGL((mult_update_kernel<double>), dim3(elem_grid(t.tn * t.c)), dim3(256), 0, ax, (double*)G,
                                   (const double*)E, (const double*)D, t.tn, t.c, (int64_t)t.c, (int64_t)t.c);
                check_launch("mult_update(rows)");
            } else {
                hipLaunchKernelGGL((mult_update_kernel<float>), dim3(elem_grid(t.tn * t.c)), dim3(256), 0, ax, (float*)G,
                                   (const float*)E, (const float*)D, t.tn, t.c, (int64_t)t.c, (int64_t)t.c);
                check_launch("mult_update(rows)");
            }
        }
        rec(ev_t(i, T_UPD), ax);
        wait(cs, ev_t(i, T_UPD));
        if (t.gather_master) collective(c, COLL_ALL_GATHER, t.G.ptr, (size_t)t.chunk * t.c, p->mt, cs);
        else collective(c, COLL_ALL_GATHER, t.Grow.ptr, (size_t)t.chunk * t.ldrow, SKF_BF16, cs);
        rec(ev_t(i, T_G), cs);
        if (p->bf16) {      // the operand copies of ALL rows from what was gathered
            wait(ax, ev_t(i, T_G));
            if (t.gather_master) refresh_gt(p, t, ax);
            else launch_to_bf16<uint16_t>((uint16_t*)t.GTb.ptr, t.ldgt, (const uint16_t*)t.Grow.ptr, t.ldrow, t.n, (int64_t)t.c, true, ax);
            rec(ev_t(i, T_GT), ax);
        }
    };
    auto side_done = [&](int type) {
        if (--sides_left[type] == 0) finish_type((size_t)type);
    };
    // W partial of relation k from its P (main stream), then its sum (exchange stream)
    auto w_partial = [&](size_t k) {
        RelState& r = p->rels[k];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        if (r.absent) {
            SKF_HIP(hipMemsetAsync(r.W.ptr, 0, r.W.bytes, st));
        } else if (r.kn) {                  // lists of known entries: the block's share of W from its stored residuals (skf_known.h)
            known_w(p, r, st);
        } else {
            GemmArgs g = gemm_args(own(ti.G, ti), 1, ti.c, r.P.ptr, tj.c, 1, r.W.ptr, tj.c, ti.c, tj.c, (int)r.nr, EPI_STORE, 0);
            wide_gemm(p, g, st);
        }
        rec(ev_r(k, R_W), st);
        wait(cs, ev_r(k, R_W));
        collective(c, COLL_ALL_REDUCE, r.W.ptr, (size_t)ti.c * tj.c, SKF_F64, cs);
        rec(ev_r(k, R_WX), cs);
    };
    // partial Q of relation k (main stream), then the rows of its sum this rank owns (exchange stream)
    auto q_partial = [&](size_t k) {
        RelState& r = p->rels[k];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        if (r.absent) SKF_HIP(hipMemsetAsync(r.Q.ptr, 0, (size_t)tj.n * ti.c * p->esz, st));
        else if (r.kn) known_pass(p, r, true, SRP_RESIDUAL, st);       // E^T G_i of the block's rows (the dense part of Q follows
        else contraction_Q(p, r, st);                                  //  the scatter, on the owned rows of the column type)
        rec(ev_r(k, R_Q), st);
        wait(cs, ev_r(k, R_Q));
        collective(c, COLL_REDUCE_SCATTER, r.Q.ptr, (size_t)tj.chunk * ti.c, p->mt, cs);
        rec(ev_r(k, R_QX), cs);
    };
    // second stream, behind the sum of W: S = K_i W K_j, its B / D terms, the rounding of S
    auto backbone_chain = [&](size_t k) {
        RelState& r = p->rels[k];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int ci = ti.c, cj = tj.c;
        wait(ax, ev_r(k, R_WX));
        bool s32_here = false;
        if (chain) {
            BackboneBatch bb;
            bb.Ki[0] = (const double*)ti.K.ptr; bb.Kj[0] = (const double*)tj.K.ptr;
            bb.W[0] = (const double*)r.W.ptr; bb.S[0] = (double*)r.S.ptr;
            bb.ci[0] = ci; bb.cj[0] = cj;
            const size_t smem = ((size_t)ci * ci + 2 * (size_t)ci * cj + (size_t)cj * cj) * 8;
            static DeviceOnce once_bb, once_bt;
            allow_dynamic_lds(once_bb, backbone_small_kernel, 4 * SMALLC * SMALLC * 8);
            allow_dynamic_lds(once_bt, bterms_small_kernel, 4 * SMALLC * SMALLC * 8);
            hipLaunchKernelGGL(backbone_small_kernel, dim3(1), dim3(256), smem, ax, bb);
            check_launch("backbone_small");
            BTermsArgs ba;
            ba.S = (const double*)r.S.ptr; ba.Gram_i = (const double*)ti.Gram.ptr; ba.Gram_j = (const double*)tj.Gram.ptr;
            ba.Bp_i = (double*)ti.Bp_tot.ptr; ba.Bn_i = (double*)ti.Bn_tot.ptr;
            ba.Bp_j = (double*)tj.Bp_tot.ptr; ba.Bn_j = (double*)tj.Bn_tot.ptr;
            ba.ci = ci; ba.cj = cj; ba.nan_to_num = nan_upd;
            hipLaunchKernelGGL(bterms_small_kernel, dim3(1), dim3(256), smem, ax, ba);
            check_launch("bterms_small");
        } else {
            GemmArgs g = gemm_args(ti.K.ptr, ci, 1, r.W.ptr, cj, 1, r.T1.ptr, cj, ci, cj, ci, EPI_STORE, 0);       // T1 = K_i W
            small_gemm(p, g, ax);
            g = gemm_args(r.T1.ptr, cj, 1, tj.K.ptr, cj, 1, r.S.ptr, cj, ci, cj, cj, EPI_STORE, 1);                // S = T1 K_j
            s32_here = !p->f64 && fused && !p->sw.no_pairs;        // f32 engines: the rounding of S leaves the same launch
            if (s32_here) {
                g.epi = EPI_STORE_F32;
                g.C2 = r.S32.ptr;
                g.ldc2 = cj;
            }
            small_gemm(p, g, ax);
            relation_small_terms(p, r, nan_upd, EPI_SPLIT_ACC, ti.Bp_tot.ptr, ti.Bn_tot.ptr, tj.Bp_tot.ptr, tj.Bn_tot.ptr,
                                 true, true, ax, r.T1.ptr);           // (T1 is free once S is there)
        }
        Sm[k] = r.S.ptr;
        if (!p->f64 && fused) {
            if (!s32_here) {
                hipLaunchKernelGGL((cast_kernel<float, double>), dim3(elem_grid((int64_t)ci * cj)), dim3(256), 0, ax,
                                   (float*)r.S32.ptr, (int64_t)cj, (const double*)r.S.ptr, (int64_t)cj, (int64_t)ci, (int64_t)cj);
                check_launch("cast");
            }
            Sm[k] = r.S32.ptr;
        }
    };
    // second stream: row side on the local rows behind `ev_p`, column side on the owned rows of the column type behind the
    // scattered Q;  E_i (+)= (P S^T)+, D_i (+)= (P S^T)-;  E_j (+)= (Q S)+, D_j (+)= (Q S)-
    auto side_products = [&](size_t k, hipEvent_t ev_p) {
        RelState& r = p->rels[k];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int ci = ti.c, cj = tj.c;
        wait(ax, ev_p);
        if (!r.absent && r.kn) {            // the row-side product P S^T = G_i (S Gram_j S^T) + E T was formed by the row pass
            known_row_split(p, r, touched[r.row] != 0, ax);
            touched[r.row] = 1;
        } else if (!r.absent) {
            if (fused) {
                side_update(p, r.P.ptr, cj, cj, Sm[k], 1, cj, ti, own(ti.G, ti), own(ti.E, ti), own(ti.D, ti), (int)r.nr, nullptr,
                            nullptr, false, touched[r.row] != 0, nan_upd, ax);
            } else {
                if (!touched[r.row]) {
                    SKF_HIP(hipMemsetAsync(own(ti.E, ti), 0, (size_t)ti.tn * ci * p->esz, ax));
                    SKF_HIP(hipMemsetAsync(own(ti.D, ti), 0, (size_t)ti.tn * ci * p->esz, ax));
                }
                GemmArgs g = gemm_args(r.P.ptr, cj, 1, r.S.ptr, 1, cj, own(ti.E, ti), ci, (int)r.nr, ci, cj, EPI_SPLIT_ACC, nan_upd);
                g.C2 = own(ti.D, ti);
                mixed_gemm(p, g, ax);
            }
            touched[r.row] = 1;
        }
        side_done(r.row);
        wait(ax, ev_r(k, R_QX));
        if (r.kn) {                         // Q = G_j (S^T Gram_i) + E^T G_i: the dense part on the owned rows, then S_prev <- S
            known_col_dense(p, r, tj.t0, tj.tn, ax, multi);
            if (!r.absent) copy2d(r.Sp.ptr, cj, r.S.ptr, cj, ci, cj, 8, ax);
        }
        if (tj.tn > 0) {
            const void* Qo = (const char*)r.Q.ptr + (size_t)tj.t0 * ci * p->esz;
            if (fused) {
                side_update(p, Qo, ci, ci, Sm[k], cj, 1, tj, own(tj.G, tj), own(tj.E, tj), own(tj.D, tj), (int)tj.tn, nullptr,
                            nullptr, false, touched[r.col] != 0, nan_upd, ax);
            } else {
                if (!touched[r.col]) {
                    SKF_HIP(hipMemsetAsync(own(tj.E, tj), 0, (size_t)tj.tn * cj * p->esz, ax));
                    SKF_HIP(hipMemsetAsync(own(tj.D, tj), 0, (size_t)tj.tn * cj * p->esz, ax));
                }
                GemmArgs g = gemm_args(Qo, ci, 1, r.S.ptr, cj, 1, own(tj.E, tj), cj, (int)tj.tn, cj, ci, EPI_SPLIT_ACC, nan_upd);
                g.C2 = own(tj.D, tj);
                mixed_gemm(p, g, ax);
            }
            touched[r.col] = 1;
        }
        side_done(r.col);
    };

    // The launches go out stream by stream, not relation by relation: every contraction of the main stream is enqueued BEFORE
    // the chains of the second stream, whose first packets wait for the pseudo-inverses (~1.2 ms at rank 256).  Streams
    // share hardware queues, and a queue hands out its packets in order: issued relation by relation, the third relation's
    // contractions sat behind the first relation's waiting chain for 0.5 ms (rocprof timeline of rank 3 of 8, config 3).
    // ---- main stream (+ exchanges): masked relations (DFMC) before their completion: P, W (_dfmc.py:311-314) ...
    for (size_t k = 0; k < nr; ++k) {
        RelState& r = p->rels[k];
        if (!(dfmc && r.masked)) continue;
        if (!r.absent && !r.kn) contraction_P(p, r, st);
        w_partial(k);
    }
    // ... and the unmasked relations: P, W, Q
    for (size_t k = 0; k < nr; ++k) {
        RelState& r = p->rels[k];
        if (dfmc && r.masked) continue;
        if (!r.absent) contraction_P(p, r, st);
        w_partial(k);
        q_partial(k);
    }
    // ---- second stream: backbone, H = G_i[own] S and the completion operands of the masked relations ...
    for (size_t k = 0; k < nr; ++k) {
        RelState& r = p->rels[k];
        if (!(dfmc && r.masked)) continue;
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        backbone_chain(k);
        if (r.kn) {                         // gathered vectors T = G_j S^T and the c x c operands of the dense parts
            if (!r.absent) {
                known_operands(p, r, ax, multi);
            } else {                        // (no rows of the relation here: only U2 = S^T Gram_i for this process's rows of Q)
                GemmArgs g = gemm_args(r.S.ptr, 1, tj.c, ti.Gram.ptr, ti.c, 1, r.U2.ptr, ti.c, tj.c, ti.c, ti.c, EPI_STORE, 0);
                small_gemm(p, g, ax);
            }
        } else if (!r.absent) {
            GemmArgs g = gemm_args(own(ti.G, ti), ti.c, 1, r.S.ptr, tj.c, 1, r.H.ptr, tj.c, (int)r.nr, tj.c, ti.c, EPI_STORE, 0);
            if (multi) mixed_gemm_unsplit(p, g, ax);
            else mixed_gemm(p, g, ax);
            if (p->bf16 && r.mask) {        // bf16 operands of the completion tiles: H, and G_j from what this rank holds of it
                launch_to_bf16<float>((uint16_t*)r.Hb.ptr, r.ldhb, (const float*)r.H.ptr, (int64_t)tj.c, r.nr, tj.c, false, ax);
                if (tj.gather_master)
                    launch_to_bf16<float>((uint16_t*)r.Gb.ptr, r.ldhb, (const float*)tj.G.ptr, (int64_t)tj.c, tj.n, tj.c, false, ax);
                else
                    launch_to_bf16<uint16_t>((uint16_t*)r.Gb.ptr, r.ldhb, (const uint16_t*)tj.Grow.ptr, tj.ldrow, tj.n, tj.c, false, ax);
            }
        }
        rec(ev_r(k, R_S), ax);
    }
    // ---- main stream: completion of the local rows of the masked relations (_dfmc.py:319-325), then the two contractions of
    // the G update
    for (size_t k = 0; k < nr; ++k) {
        RelState& r = p->rels[k];
        if (!(dfmc && r.masked)) continue;
        TypeState& tj = p->types[r.col];
        wait(st, ev_r(k, R_S));
        if (r.kn) {                         // the row pass stands for completion + P: A = E T + G_i (S Gram_j S^T)
            if (!r.absent) {
                known_row_pass(p, r, st);
                known_row_dense(p, r, st, false);
            }
        } else if (!r.absent) {
            if (r.mask) {
                if (p->bf16) {
                    launch_tile_epilogue(p, r, MODE_COMPLETE, st, false);
                } else {
                    GemmArgs g = gemm_args(r.H.ptr, tj.c, 1, tj.G.ptr, 1, tj.c, r.Rw.ptr, r.ldr, (int)r.nr, (int)tj.n, tj.c,
                                           EPI_MASKED_STORE, 0);
                    g.mask = (const uint8_t*)r.Mb.ptr;
                    g.ldmask = r.ldmb;
                    g.mask_bits = 1;
                    plan_gemm(p, g, st);
                }
            }
            contraction_P(p, r, st);
        }
        rec(ev_r(k, R_P2), st);
        q_partial(k);
    }
    // ---- second stream: backbones and side products of the unmasked relations, then the side products of the masked ones
    for (size_t k = 0; k < nr; ++k) {
        RelState& r = p->rels[k];
        if (dfmc && r.masked) continue;
        backbone_chain(k);
        side_products(k, ev_r(k, R_WX));
    }
    for (size_t k = 0; k < nr; ++k) {
        RelState& r = p->rels[k];
        if (!(dfmc && r.masked)) continue;
        side_products(k, ev_r(k, R_P2));
    }
    for (size_t i = 0; i < nt; ++i)
        if (!finished[i]) finish_type(i);          // a type without relations: constraints only
    // ---- the next iteration starts behind every gather (and operand refresh)
    for (size_t i = 0; i < nt; ++i) wait(st, ev_t(i, p->bf16 ? T_GT : T_G));
    p->masters_stale = false;
    for (const TypeState& t : p->types) p->masters_stale = p->masters_stale || !t.gather_master;
    p->kn_first = false;
}

// SKF_BF16 plans with owned rows gather only the bf16 operand rows of a type without constraints; at the end of
// skf_iterate_dist every rank fetches the f32 rows of the other owners once, so that skf_get_factor / skf_relation_sqerr see
// the same factors everywhere.
static void finalize_owned(skf_plan* p, hipStream_t st) {
    if (!p->masters_stale) return;
    // on the stream the iteration's exchanges went out on (one communicator, one order of collectives on every rank)
    hipStream_t cs = (owned_can_overlap(p) && p->cs) ? p->cs : st;
    if (cs != st) {
        SKF_HIP(hipEventRecord(p->ev_own[0], st));
        SKF_HIP(hipStreamWaitEvent(cs, p->ev_own[0], 0));
    }
    for (TypeState& t : p->types)
        if (!t.gather_master) collective(p->comm, COLL_ALL_GATHER, t.G.ptr, (size_t)t.chunk * t.c, p->mt, cs);
    if (cs != st) {
        SKF_HIP(hipEventRecord(p->ev_own[1], cs));
        SKF_HIP(hipStreamWaitEvent(st, p->ev_own[1], 0));
    }
    p->masters_stale = false;
}

// ------------------------------------------------------------------------------------------
// The same DFMF iteration as stage_contract + stage_backbone + stage_accumulate, scheduled as a pipeline over
// the relations (whole, unmasked relations; every rank between 65 and 512; MFMA engine): the main stream runs
// nothing but the Gram products and the contractions P_r, Q_r, W_r; everything that follows a relation's
// contractions -- S_r = K_i W_r K_j, the B / D terms, the roundings, its two side products -- runs on the second
// stream UNDERNEATH the contractions of the next relation (those launches leave 60 of the 256 CUs idle when their
// 196 row tiles run unsplit, and every launch has a tail).  Only the last relation's small work, the type-level
// terms G (sum B) and the update itself remain exposed.  Arithmetic and results are those of the staged schedule;
// only the order in which the relations add into E / D differs (relations are walked cheapest-last).
// ------------------------------------------------------------------------------------------
static bool can_pipeline(const skf_plan* p) {
    if (!p->overlap || p->sliced || p->engine != SKF_ENGINE_MFMA || !p->pipeline) return false;
    if (p->variant != SKF_DFMF && p->variant != SKF_DFMC) return false;
    if (p->rels.empty() || p->rels.size() > 64) return false;
    for (const ThetaState& th : p->thetas)
        if (!th.sparse) return false;            // (dense constraint products share the split-K scratch of the main stream)
    // every rank <= 64: the one-workgroup chains of the staged schedule are the faster path (launch-bound graphs);
    // above 512 (DFMC: 256) a c x c product on the second stream could ask for the main stream's split-K scratch
    const int cmax = (p->variant == SKF_DFMC) ? 256 : 512;
    bool all_small = true;
    for (const TypeState& t : p->types) {
        if (t.c > cmax) return false;
        if (t.c > SMALLC) all_small = false;
    }
    if (all_small) return false;
    if (p->variant == SKF_DFMF)
        for (const TypeState& t : p->types)
            if (t.c <= SMALLC) return false;     // (DFMF: the mixed case stays on the schedule it was measured on)
    for (const RelState& r : p->rels)
        if (r.absent || (r.masked && p->variant != SKF_DFMC)) return false;
    return true;
}

// DFMC adds one dependency to the pipeline: a masked relation is contracted once BEFORE its completion (W = G_i^T R G_j
// of the backbone, _dfmc.py:311-314) and twice after it (_dfmc.py:319-325, 341-345).  Main stream: Gram products; the
// first contraction and W of every masked relation; P, W, Q of the unmasked relations; then completion, P, Q of the
// masked ones -- by then their backbones and reconstruction operands, computed on the second stream underneath the
// unmasked relations' contractions, are long done.
static void iterate_fit_pipelined(skf_plan* p, hipStream_t st) {
    const size_t nt = p->types.size(), nr = p->rels.size();
    hipStream_t ax = p->aux;
    const bool dfmc = (p->variant == SKF_DFMC);
    const int nan_upd = dfmc ? 0 : 1;             // DFMF: nan_to_num on the A/B/C/D terms (_dfmf.py:254-276); DFMC: none (_dfmc.py:127-178)
    if (dfmc && p->first_iter) zero_unknown_entries(p, st);
    p->first_iter = false;
    // per relation: [0] first contraction and W done (main) | [1] backbone and reconstruction operands done (second
    // stream, masked relations) | [2] P done (main) | [3] Q done (main)
    if (p->ev_rel.size() < 4 * nr) {
        const size_t old = p->ev_rel.size();
        p->ev_rel.resize(4 * nr);
        for (size_t k = old; k < 4 * nr; ++k) SKF_HIP(hipEventCreateWithFlags(&p->ev_rel[k], hipEventDisableTiming));
    }
    // order of the relations: most expensive first, so that the exposed tail belongs to the cheapest one
    std::vector<size_t> order(nr);
    for (size_t k = 0; k < nr; ++k) order[k] = k;
    auto cost = [&](size_t k) {
        const RelState& r = p->rels[k];
        return (double)r.nr * (double)p->types[r.col].n * (p->types[r.row].c + p->types[r.col].c);
    };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost(a) > cost(b); });

    // ---- main: Gram matrices; second stream: their pseudo-inverses, then the cleared B sums.
    // (Measured, profiles/r02_pipeline_ab.txt: with the Gram products and W = G_i^T P on the second stream as well the
    // contractions slow down by as much as those kernels take -- they are whole-chip launches and any co-resident
    // workgroup takes a CU from them; what the pipeline can hide is what fits on the 60 CUs the unsplit 196-tile
    // launches leave idle and in the launch tails: +1.5 % it/s.)
    // DFMC with a relation kept as lists of its known entries: the main stream opens with the pass over the stored
    // residuals, which is bound by its gathers and leaves the matrix cores idle -- there the Gram products go to the second
    // stream and run BESIDE the pass (config 5: 138.0 -> 140.9 it/s; with the cross-Gram matrices of W there as well, ahead
    // of the pseudo-inverses: 139.1 -- they hold up the chain the row pass waits for; profiles/r03_c5_overlap_ab.txt).
    std::vector<int> all;
    bool any_kn = false;
    for (const RelState& r : p->rels) any_kn = any_kn || r.kn;
    const bool gram_aux = dfmc && any_kn;
    if (gram_aux) {
        SKF_HIP(hipEventRecord(p->ev_fork, st));
        SKF_HIP(hipStreamWaitEvent(ax, p->ev_fork, 0));
    }
    for (size_t i = 0; i < nt; ++i) {
        gram(p, p->types[i], 1, gram_aux ? ax : st, gram_aux);
        all.push_back((int)i);
    }
    if (!gram_aux) {
        SKF_HIP(hipEventRecord(p->ev_fork, st));
        SKF_HIP(hipStreamWaitEvent(ax, p->ev_fork, 0));
    }
    // (Round 5, measured and not kept: for DFMC on known-entry lists the pseudo-inverses on the MAIN stream behind the first
    // relation's pass -- here they take 0.78 ms underneath the other relations' passes, 0.2 ms alone; config 5 147.9 against
    // 149.3 it/s: the passes they hold up are worth more than the chains they release.)
    plan_pinv(p, all, ax);
    SKF_HIP(hipMemsetAsync((char*)p->ws_base + p->btot_off, 0, p->btot_bytes, ax));

    std::vector<char> touched(nt, 0);
    std::vector<int> rels_left(nt, 0);           // relations of the type still to come: its sum of B is complete at 0
    for (const RelState& r : p->rels) {
        rels_left[r.row] += 1;
        rels_left[r.col] += 1;
    }
    std::vector<char> updated(nt, 0), has_theta(nt, 0);       // early updates (below): DFMF only, types without constraints
    for (const ThetaState& th : p->thetas) has_theta[th.type] = 1;
    const bool early_ok = !dfmc && p->sw.early_update;
    // type-level terms E_i += G_i sum B-, D_i += G_i sum B+ as soon as the last relation of the type is through
    auto type_term = [&](size_t i) {
        TypeState& t = p->types[i];
        const void* Bn = t.Bn_tot.ptr;
        const void* Bp = t.Bp_tot.ptr;
        if (!p->f64) {
            cast_b_sums(t, ax);
            Bn = t.Bn32.ptr;
            Bp = t.Bp32.ptr;
        }
        if (!touched[i]) {                        // a type without relations here: E = D = 0 before the term is added
            SKF_HIP(hipMemsetAsync(t.E.ptr, 0, t.E.bytes, ax));
            SKF_HIP(hipMemsetAsync(t.D.ptr, 0, t.D.bytes, ax));
            touched[i] = 1;
        }
        side_update(p, nullptr, 0, 0, nullptr, 0, 0, t, t.G.ptr, t.E.ptr, t.D.ptr, (int)t.n, Bn, Bp, true, true, 0, ax);
    };
    std::vector<const void*> Sm(nr, nullptr);     // the backbone in the master type, per position in `order`
    // second stream, behind event [0] of the relation: S = K_i W K_j, its B / D terms, the rounding of S
    auto backbone_chain = [&](size_t q) {
        RelState& r = p->rels[order[q]];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int ci = ti.c, cj = tj.c;
        SKF_HIP(hipStreamWaitEvent(ax, p->ev_rel[4 * q], 0));
        GemmArgs g = gemm_args(ti.K.ptr, ci, 1, r.W.ptr, cj, 1, r.T1.ptr, cj, ci, cj, ci, EPI_STORE, 0);       // T1 = K_i W
        small_gemm(p, g, ax);
        g = gemm_args(r.T1.ptr, cj, 1, tj.K.ptr, cj, 1, r.S.ptr, cj, ci, cj, cj, EPI_STORE, 1);                // S = T1 K_j
        const bool s32_here = !p->f64 && !p->sw.no_pairs;          // f32 engines: the rounding of S leaves the same launch
        if (s32_here) {
            g.epi = EPI_STORE_F32;
            g.C2 = r.S32.ptr;
            g.ldc2 = cj;
        }
        small_gemm(p, g, ax);
        relation_small_terms(p, r, nan_upd, EPI_SPLIT_ACC, ti.Bp_tot.ptr, ti.Bn_tot.ptr, tj.Bp_tot.ptr, tj.Bn_tot.ptr,
                             true, true, ax, r.T1.ptr);               // (T1 is free once S is there)
        Sm[q] = r.S.ptr;
        if (!p->f64) {
            if (!s32_here) {
                hipLaunchKernelGGL((cast_kernel<float, double>), dim3(elem_grid((int64_t)ci * cj)), dim3(256), 0, ax,
                                   (float*)r.S32.ptr, (int64_t)cj, (const double*)r.S.ptr, (int64_t)cj, (int64_t)ci, (int64_t)cj);
                check_launch("cast");
            }
            Sm[q] = r.S32.ptr;
        }
    };
    // second stream: the two side products of the relation behind `ev_p` (P) and event [3] (Q), then the type terms
    // row side: E_i (+)= (P S^T)+, D_i (+)= (P S^T)-;  column side: E_j (+)= (Q S)+, D_j (+)= (Q S)-
    auto side_products = [&](size_t q, size_t ev_p) {
        RelState& r = p->rels[order[q]];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int ni = (int)r.nr, nj = (int)tj.n, ci = ti.c, cj = tj.c;
        SKF_HIP(hipStreamWaitEvent(ax, p->ev_rel[ev_p], 0));
        if (r.kn)
            known_row_split(p, r, touched[r.row] != 0, ax);
        else
            side_update(p, r.P.ptr, cj, cj, Sm[q], 1, cj, ti, ti.G.ptr, ti.E.ptr, ti.D.ptr, ni, nullptr, nullptr, false,
                        touched[r.row] != 0, nan_upd, ax);
        touched[r.row] = 1;
        if (--rels_left[r.row] == 0) type_term(r.row);            // (before the wait for Q: under the relation's own Q)
        // the type-level term of the column type needs the B sums only (complete with this relation's backbone), not Q:
        // it goes out under the relation's own Q as well, and only the column side product remains behind Q
        const bool last_col = --rels_left[r.col] == 0;
        if (last_col) type_term(r.col);
        SKF_HIP(hipStreamWaitEvent(ax, p->ev_rel[4 * q + 3], 0));
        side_update(p, r.Q.ptr, ci, ci, Sm[q], cj, 1, tj, tj.G.ptr, tj.E.ptr, tj.D.ptr, nj, nullptr, nullptr, false,
                    touched[r.col] != 0, nan_upd, ax);
        touched[r.col] = 1;
    };
    auto w_product = [&](RelState& r, bool by_q) {        // W = G_i^T P, or (R^T G_i)^T G_j through the narrower factor
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        GemmArgs g = by_q ? gemm_args(r.Q.ptr, 1, ti.c, tj.G.ptr, tj.c, 1, r.W.ptr, tj.c, ti.c, tj.c, (int)tj.n, EPI_STORE, 0)
                          : gemm_args(ti.G.ptr, 1, ti.c, r.P.ptr, tj.c, 1, r.W.ptr, tj.c, ti.c, tj.c, (int)r.nr, EPI_STORE, 0);
        wide_gemm(p, g, st);
    };

    // ---- masked relations (DFMC), before their completion: the contraction W needs; second stream: backbone, H = G_i S
    for (size_t q = 0; q < nr; ++q) {
        RelState& r = p->rels[order[q]];
        if (!(dfmc && r.masked)) continue;
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const bool by_q = ti.c < tj.c;
        if (r.kn) {           // known entries only: W from the stored residuals; second stream: backbone, gathered vectors, c x c operands
            // (Measured, profiles/r03_c5_known_entries.txt: on a third stream, beside the matrix-core contractions of the
            // unmasked relations, the pass over the stored residuals just time-slices the chip with them -- 2.0 ms instead
            // of 0.55 ms for user x tag, 92.2 vs 91.6 it/s.  Its 10 000 workgroups leave no CU to share.)
            known_w(p, r, st);
            SKF_HIP(hipEventRecord(p->ev_rel[4 * q], st));
            backbone_chain(q);
            known_operands(p, r, ax, true);
            SKF_HIP(hipEventRecord(p->ev_rel[4 * q + 1], ax));
            continue;
        }
        if (by_q) contraction_Q(p, r, st);
        else contraction_P(p, r, st);
        w_product(r, by_q);
        SKF_HIP(hipEventRecord(p->ev_rel[4 * q], st));
        backbone_chain(q);
        GemmArgs g = gemm_args(ti.G.ptr, ti.c, 1, r.S.ptr, tj.c, 1, r.H.ptr, tj.c, (int)r.nr, tj.c, ti.c, EPI_STORE, 0);
        mixed_gemm_unsplit(p, g, ax);
        if (p->bf16 && r.mask) tile_epilogue_operands(p, r, ax);
        SKF_HIP(hipEventRecord(p->ev_rel[4 * q + 1], ax));
    }
    // ---- unmasked relations: P, W, Q on the main stream.  W goes out between P and Q: the backbone chain and the row
    // side of the relation then run underneath its OWN Q (for the last relation the exposed tail is the column side only).
    // (DFMC: putting the cheap half of them behind the completion block, to shorten the exposed tail, measured equal --
    // their chains are latency-bound and hide better under the large launches; profiles/r02_pipeline_ab.txt.)
    for (size_t q = 0; q < nr; ++q) {
        RelState& r = p->rels[order[q]];
        if (dfmc && r.masked) continue;
        // W through the shorter of the two object dimensions: G_i^T P sums over the rows of the relation, Q^T G_j over its
        // columns (config 3, 100k x 40k: 0.32 -> 0.13 ms).  The Q form has to wait for Q, so the relation's chain then runs
        // under the NEXT relation's contractions instead of its own Q -- not for the last relation, whose tail it would grow.
        const bool w_by_q = (q + 1 < nr) && 5 * p->types[r.col].n <= 3 * r.nr;
        contraction_P(p, r, st);
        if (!w_by_q) {
            w_product(r, false);
            SKF_HIP(hipEventRecord(p->ev_rel[4 * q], st));
        }
        contraction_Q(p, r, st);
        SKF_HIP(hipEventRecord(p->ev_rel[4 * q + 3], st));
        if (w_by_q) {
            w_product(r, true);
            SKF_HIP(hipEventRecord(p->ev_rel[4 * q], st));
        }
        backbone_chain(q);
        side_products(q, 4 * q);
        // a type whose last relation this was has its E and D complete behind the side products just issued, and nothing
        // that is still to come reads its factor (no later relation, no constraint on it): its update -- for the bf16 engine
        // also the refresh of the stored G^T -- goes out on the second stream NOW, underneath the remaining relations'
        // contractions, instead of at the exposed end of the iteration (config 3: the 100k-object type, 88 of the 150 us
        // the three updates take).  Same arithmetic, same results.
        if (early_ok)
            for (int side = 0; side < 2; ++side) {
                const size_t i = side == 0 ? (size_t)r.row : (size_t)r.col;
                if (rels_left[i] == 0 && !updated[i] && !has_theta[i] && q + 1 < nr) {
                    update_type(p, p->types[i], ax);
                    updated[i] = 1;
                }
            }
    }
    // ---- masked relations: completion (_dfmc.py:319-325), then the two contractions of the G update
    for (size_t q = 0; q < nr; ++q) {
        RelState& r = p->rels[order[q]];
        if (!(dfmc && r.masked)) continue;
        TypeState& tj = p->types[r.col];
        SKF_HIP(hipStreamWaitEvent(st, p->ev_rel[4 * q + 1], 0));
        if (r.kn) {           // the two residual passes over the lists stand for completion, P and Q
            known_row_pass(p, r, st);
            known_row_dense(p, r, st, false);       // (0.1 ms here; on the low-priority second stream it crawled for 2 ms
                                                    // underneath the column pass and slowed that pass down by 0.4 ms)
            SKF_HIP(hipEventRecord(p->ev_rel[4 * q + 2], st));
            known_col_pass(p, r, st);
            SKF_HIP(hipEventRecord(p->ev_rel[4 * q + 3], st));
            side_products(q, 4 * q + 2);
            continue;
        }
        if (r.mask) {
            if (p->bf16) {
                launch_tile_epilogue(p, r, MODE_COMPLETE, st, false);
            } else {
                GemmArgs g = gemm_args(r.H.ptr, tj.c, 1, tj.G.ptr, 1, tj.c, r.Rw.ptr, r.ldr, (int)r.nr, (int)tj.n, tj.c,
                                       EPI_MASKED_STORE, 0);
                g.mask = (const uint8_t*)r.Mb.ptr;
                g.ldmask = r.ldmb;
                g.mask_bits = 1;
                plan_gemm(p, g, st);
            }
        }
        contraction_P(p, r, st);
        SKF_HIP(hipEventRecord(p->ev_rel[4 * q + 2], st));
        contraction_Q(p, r, st);
        SKF_HIP(hipEventRecord(p->ev_rel[4 * q + 3], st));
        side_products(q, 4 * q + 2);
    }
    for (size_t i = 0; i < nt; ++i)
        if (rels_left[i] == 0 && !touched[i]) type_term(i);          // types without relations in this plan
    theta_terms(p, ax);
    SKF_HIP(hipEventRecord(p->ev_join, ax));
    SKF_HIP(hipStreamWaitEvent(st, p->ev_join, 0));
    apply_update(p, st, &updated);
}

// The DFMF iteration of a small graph in three launches (skf_small.h); n_batch > 1: of that many plans of the same graph at
// once -- independent restarts side by side, blockIdx.y = the plan (skf_iterate_batch; p->sm_batch holds their tables)
template <typename T>
static void iterate_small_fused_t(skf_plan* p, hipStream_t st, unsigned n_batch = 1) {
    const SmTables* tb = (const SmTables*)p->sm_tables.ptr;
    const SmTables* const* tbs = (const SmTables* const*)p->sm_batch.ptr;
    constexpr int bb_lds = (2 * 64 + 2 * SM_BK) * SM_LD * 8;
    const SmJob* j1 = (const SmJob*)p->sm_jobs1.ptr;
    const SmJob* j3 = (const SmJob*)p->sm_jobs3.ptr;
    const unsigned n1 = (unsigned)p->sm_j1.size(), n2 = (unsigned)(2 * p->rels.size()), n3 = (unsigned)p->sm_j3.size();
    if (n_batch > 1) {          // restarts side by side: blockIdx.y = the plan
        static DeviceOnce once_c, once_b, once_u;
        allow_dynamic_lds(once_c, small_contract_kernel<T, true>, SM_TILE_BYTES);
        allow_dynamic_lds(once_b, small_backbone_kernel<true>, bb_lds);
        allow_dynamic_lds(once_u, small_update_kernel<T, true>, SM_TILE3_BYTES);
        hipLaunchKernelGGL((small_contract_kernel<T, true>), dim3(n1, n_batch), dim3(256), SM_TILE_BYTES, st, tb, tbs, j1);
        check_launch("small_contract");
        hipLaunchKernelGGL((small_backbone_kernel<true>), dim3(n2, n_batch), dim3(256), bb_lds, st, tb, tbs);
        check_launch("small_backbone");
        hipLaunchKernelGGL((small_update_kernel<T, true>), dim3(n3, n_batch), dim3(256), SM_TILE3_BYTES, st, tb, tbs, j3);
        check_launch("small_update");
    } else {                    // one plan: its tables are the kernel argument
        static DeviceOnce once_c, once_b, once_u;
        allow_dynamic_lds(once_c, small_contract_kernel<T, false>, SM_TILE_BYTES);
        allow_dynamic_lds(once_b, small_backbone_kernel<false>, bb_lds);
        allow_dynamic_lds(once_u, small_update_kernel<T, false>, SM_TILE3_BYTES);
        hipLaunchKernelGGL((small_contract_kernel<T, false>), dim3(n1), dim3(256), SM_TILE_BYTES, st, tb, tbs, j1);
        check_launch("small_contract");
        hipLaunchKernelGGL((small_backbone_kernel<false>), dim3(n2), dim3(256), bb_lds, st, tb, tbs);
        check_launch("small_backbone");
        hipLaunchKernelGGL((small_update_kernel<T, false>), dim3(n3), dim3(256), SM_TILE3_BYTES, st, tb, tbs, j3);
        check_launch("small_update");
    }
    p->first_iter = false;
}

static void iterate_fit(skf_plan* p, hipStream_t st) {
    if (p->small_fused) {
        if (p->f64) iterate_small_fused_t<double>(p, st);
        else iterate_small_fused_t<float>(p, st);
        return;
    }
    if (can_pipeline(p)) {
        iterate_fit_pipelined(p, st);
        return;
    }
    accumulate_fit(p, st);
    apply_update(p, st);
}

static bool use_graph(skf_plan* p, hipStream_t st, int n_iters) {
    if (st == nullptr || p->profiling || p->graph_failed || n_iters < 4) return false;
    // opt-in (SKF_GRAPH=1): measured on dicty (50 launches / 0.5 ms iteration) the replay is not
    // faster than the asynchronous eager launches -- the iteration is bound by kernel time
    return p->graph_on || p->sw.graph;
}

// Record one iteration into a hipGraph (the second-stream fork/join becomes graph edges).
// Any failure leaves the plan on the eager path.
static void capture_iteration(skf_plan* p, hipStream_t st) {
    if (p->graph_exec) {
        (void)hipGraphExecDestroy(p->graph_exec);
        p->graph_exec = nullptr;
    }
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        p->graph_failed = true;
        return;
    }
    hipGraph_t graph = nullptr;
    bool ok = true;
    try {
        iterate_fit(p, st);
    } catch (const Error&) {
        ok = false;
    }
    if (hipStreamEndCapture(st, &graph) != hipSuccess || !ok || graph == nullptr) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        p->graph_failed = true;
        return;
    }
    if (hipGraphInstantiate(&p->graph_exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        p->graph_exec = nullptr;
        p->graph_failed = true;
    }
    (void)hipGraphDestroy(graph);
    p->graph_stream = st;
}

// SKF_TRANSFORM: everything that does not depend on G_target is computed once.
static void prepare_transform(skf_plan* p, hipStream_t st) {
    TypeState& tt = p->types[p->target];
    SKF_HIP(hipMemsetAsync(tt.Ec.ptr, 0, tt.Ec.bytes, st));
    SKF_HIP(hipMemsetAsync(tt.Dc.ptr, 0, tt.Dc.bytes, st));
    SKF_HIP(hipMemsetAsync(tt.Bp_tot.ptr, 0, tt.Bp_tot.bytes, st));
    SKF_HIP(hipMemsetAsync(tt.Bn_tot.ptr, 0, tt.Bn_tot.bytes, st));
    for (size_t i = 0; i < p->types.size(); ++i)
        if ((int)i != p->target) gram(p, p->types[i], 0, st);
    for (RelState& r : p->rels) {
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int ni = (int)ti.n, nj = (int)tj.n, ci = ti.c, cj = tj.c;
        if (r.row == p->target) {             // _dfmf.py:392-405
            GemmArgs g = gemm_args(r.R, r.ldr, 1, tj.G.ptr, cj, 1, r.P.ptr, cj, ni, cj, nj, EPI_STORE, 0);
            relation_gemm(p, g, st, &r, false);
            g = gemm_args(r.P.ptr, cj, 1, r.S.ptr, 1, cj, tt.Ec.ptr, ci, ni, ci, cj, EPI_SPLIT_ACC, 0);
            g.C2 = tt.Dc.ptr;
            mixed_gemm(p, g, st);
            relation_small_terms(p, r, 0, EPI_SPLIT_ACC, tt.Bp_tot.ptr, tt.Bn_tot.ptr, nullptr, nullptr, true,
                                 false, st);
        } else {                              // _dfmf.py:407-419
            GemmArgs g = gemm_args(r.R, 1, r.ldr, ti.G.ptr, ci, 1, r.Q.ptr, ci, nj, ci, ni, EPI_STORE, 0);
            relation_gemm(p, g, st, &r, true);
            g = gemm_args(r.Q.ptr, ci, 1, r.S.ptr, cj, 1, tt.Ec.ptr, cj, nj, cj, ci, EPI_SPLIT_ACC, 0);
            g.C2 = tt.Dc.ptr;
            mixed_gemm(p, g, st);
            relation_small_terms(p, r, 0, EPI_SPLIT_ACC, nullptr, nullptr, tt.Bp_tot.ptr, tt.Bn_tot.ptr, false,
                                 true, st);
        }
    }
    p->prepared = true;
}

// The fold-in iteration as ONE launch (foldin_step_kernel): no constraint on the target type (a constraint couples the
// rows of an iteration through E / D, which this form never writes)
static bool fold_fused(const skf_plan* p) {
    return p->variant == SKF_TRANSFORM && p->engine == SKF_ENGINE_MFMA && p->thetas.empty();
}

// `iters` fold-in iterations of `n_plans` plans of one graph (same object count and rank of the target), every launch
// serving all of them; each plan's factor ends in its G slot (the two buffers swap roles per launch)
static void fold_steps(skf_plan* const* plans, int n_plans, int iters, hipStream_t st) {
    skf_plan* p0 = plans[0];
    const TypeState& t0 = p0->types[p0->target];
    for (int b0 = 0; b0 < n_plans; b0 += FOLD_MAXB) {
        const int nb = n_plans - b0 < FOLD_MAXB ? n_plans - b0 : FOLD_MAXB;
        for (int it = 0; it < iters; ++it) {
            FoldArgs a;
            memset(&a, 0, sizeof a);
            a.n = (int)t0.n; a.c = t0.c;
            for (int k = 0; k < nb; ++k) {
                TypeState& t = plans[b0 + k]->types[plans[b0 + k]->target];
                a.G[k] = t.G.ptr; a.Gout[k] = t.Galt.ptr;
                a.Bn[k] = t.Bn_tot.ptr; a.Bp[k] = t.Bp_tot.ptr;
                a.Ec[k] = t.Ec.ptr; a.Dc[k] = t.Dc.ptr;
            }
            if (p0->f64) {
                if (t0.n > 64 && t0.c > 64) {
                    dim3 grid(cdiv(t0.c, 64), cdiv(t0.n, 64), nb);
                    hipLaunchKernelGGL((foldin_step_kernel<double, 2, 2, 16>), grid, dim3(GEMM_THREADS), 0, st, a);
                } else {
                    dim3 grid(cdiv(t0.c, 32), cdiv(t0.n, 32), nb);
                    hipLaunchKernelGGL((foldin_step_kernel<double, 1, 1, 16>), grid, dim3(GEMM_THREADS), 0, st, a);
                }
            } else {
                dim3 grid(cdiv(t0.c, 64), cdiv(t0.n, 64), nb);
                hipLaunchKernelGGL((foldin_step_kernel<float, 1, 1, 16>), grid, dim3(GEMM_THREADS), 0, st, a);
            }
            check_launch("foldin_step");
            for (int k = 0; k < nb; ++k) {
                TypeState& t = plans[b0 + k]->types[plans[b0 + k]->target];
                std::swap(t.G.ptr, t.Galt.ptr);
                std::swap(t.G.off, t.Galt.off);
            }
        }
    }
}

static void iterate_transform(skf_plan* p, hipStream_t st) {
    if (fold_fused(p)) {
        skf_plan* one[1] = {p};
        fold_steps(one, 1, 1, st);
        return;
    }
    TypeState& tt = p->types[p->target];
    const int n = (int)tt.n, c = tt.c;
    SKF_HIP(hipMemcpyAsync(tt.E.ptr, tt.Ec.ptr, tt.E.bytes, hipMemcpyDeviceToDevice, st));
    SKF_HIP(hipMemcpyAsync(tt.D.ptr, tt.Dc.ptr, tt.D.bytes, hipMemcpyDeviceToDevice, st));
    GemmArgs g = gemm_args(tt.G.ptr, c, 1, tt.Bn_tot.ptr, c, 1, tt.E.ptr, c, n, c, c, EPI_ACC, 0);
    mixed_gemm(p, g, st);
    g = gemm_args(tt.G.ptr, c, 1, tt.Bp_tot.ptr, c, 1, tt.D.ptr, c, n, c, c, EPI_ACC, 0);
    mixed_gemm(p, g, st);
    theta_terms(p, st);
    mult_update(p, tt, st);
    if (!p->thetas.empty()) refresh_gt(p, tt, st);      // SKF_BF16: the constraint products read the stored G^T
}

// The known entries of a masked relation as row lists and column lists (bind time): counts per (row, column part) on the
// device, prefix sums on the host, fills on the device; the column lists are the transpose of the row lists, every
// (column, row part) segment sorted by row.  R values come from the caller's relation, which is not referenced afterwards.
template <typename TR, typename TM>
static void build_known_lists_t(skf_plan* p, RelState& r, hipStream_t st) {
    const int64_t rows = r.nr, cols = p->types[r.col].n;
    const int pc = r.kn_pc, pr = r.kn_pr;
    const int wgrid = (int)((rows + 3) / 4 < 2048 ? ((rows + 3) / 4 > 0 ? (rows + 3) / 4 : 1) : 2048);
    int* cnt = (int*)r.KCnt.ptr;
    hipLaunchKernelGGL(known_row_count_kernel, dim3(wgrid), dim3(256), 0, st, (const uint8_t*)r.Mb.ptr, r.ldmb, rows, cols, pc,
                       r.kn_pw, cnt);
    check_launch("known_row_count");
    const size_t nseg_r = (size_t)rows * pc, nseg_c = (size_t)cols * pr;
    std::vector<int> hc(std::max(nseg_r, nseg_c));
    std::vector<int64_t> hp(std::max(nseg_r, nseg_c) + 1);
    SKF_HIP(hipMemcpyAsync(hc.data(), cnt, nseg_r * 4, hipMemcpyDeviceToHost, st));
    SKF_HIP(hipStreamSynchronize(st));
    int64_t tot = 0;
    for (size_t k = 0; k < nseg_r; ++k) { hp[k] = tot; tot += hc[k]; }
    hp[nseg_r] = tot;
    if (tot > r.kn_cap)
        SKF_FAIL(SKF_E_INVALID, "a masked relation holds %lld known entries, more than the bound %lld given in skf_relation_desc.known_bound",
                 (long long)tot, (long long)r.kn_cap);
    r.kn_nnz = tot;
    SKF_HIP(hipMemcpyAsync(r.KrPtr.ptr, hp.data(), (nseg_r + 1) * 8, hipMemcpyHostToDevice, st));
    SKF_HIP(hipMemsetAsync(cnt, 0, 2 * nseg_c * 4, st));
    SKF_HIP(hipStreamSynchronize(st));                      // (`hp` is reused below)
    int* fillpos = cnt + nseg_c;
    if (tot > 0) {
        hipLaunchKernelGGL((known_row_fill_kernel<TR, TM>), dim3(wgrid), dim3(256), 0, st, (const uint8_t*)r.Mb.ptr, r.ldmb, rows, cols,
                           pc, (const int64_t*)r.KrPtr.ptr, (const TR*)r.R_in, r.ld_in, (int*)r.KrIdx.ptr, (TM*)r.KrVal.ptr);
        hipLaunchKernelGGL(known_col_count_kernel, dim3(wgrid), dim3(256), 0, st, (const int64_t*)r.KrPtr.ptr, (const int*)r.KrIdx.ptr,
                           pc, rows, pr, r.kn_ph, cnt);
        check_launch("known_row_fill");
    }
    SKF_HIP(hipMemcpyAsync(hc.data(), cnt, nseg_c * 4, hipMemcpyDeviceToHost, st));
    SKF_HIP(hipStreamSynchronize(st));
    int64_t t2 = 0;
    for (size_t k = 0; k < nseg_c; ++k) { hp[k] = t2; t2 += hc[k]; }
    hp[nseg_c] = t2;
    SKF_HIP(hipMemcpyAsync(r.KcPtr.ptr, hp.data(), (nseg_c + 1) * 8, hipMemcpyHostToDevice, st));
    if (tot > 0) {
        hipLaunchKernelGGL(known_col_fill_kernel, dim3(wgrid), dim3(256), 0, st, (const int64_t*)r.KrPtr.ptr, (const int*)r.KrIdx.ptr,
                           pc, rows, pr, r.kn_ph, (const int64_t*)r.KcPtr.ptr, fillpos, (int*)r.KcIdx.ptr);
        hipLaunchKernelGGL(csc_sort_kernel, dim3(elem_grid((int64_t)nseg_c)), dim3(256), 0, st, (const int64_t*)r.KcPtr.ptr,
                           (int*)r.KcIdx.ptr, (int64_t)nseg_c);
        const int cgrid = (int)((cols + 3) / 4 < 2048 ? ((cols + 3) / 4 > 0 ? (cols + 3) / 4 : 1) : 2048);
        hipLaunchKernelGGL((known_col_values_kernel<TR, TM>), dim3(cgrid), dim3(256), 0, st, (const int64_t*)r.KcPtr.ptr,
                           (const int*)r.KcIdx.ptr, pr, cols, (const TR*)r.R_in, r.ld_in, (TM*)r.KcVal.ptr);
        check_launch("known_col_fill");
        // before the first iteration the completed relation is the known entries and zeros (_dfmc.py:287-292): E = R there
        SKF_HIP(hipMemcpyAsync(r.KcE.ptr, r.KcVal.ptr, (size_t)tot * sizeof(TM), hipMemcpyDeviceToDevice, st));
    }
    SKF_HIP(hipMemsetAsync(r.Sp.ptr, 0, r.Sp.bytes, st));
    if (r.FiB.bytes) SKF_HIP(hipMemsetAsync(r.FiB.ptr, 0, r.FiB.bytes, st));
    SKF_HIP(hipStreamSynchronize(st));                      // the host vectors die here; bind is not on the hot path
    r.R = nullptr;                                          // nothing reads the relation itself after this
}
static void build_known_lists(skf_plan* p, RelState& r, hipStream_t st) {
    if (p->bf16) build_known_lists_t<uint16_t, float>(p, r, st);
    else if (p->f64) build_known_lists_t<double, double>(p, r, st);
    else build_known_lists_t<float, float>(p, r, st);
}

}  // namespace skf

using namespace skf;

extern "C" {

const char* skf_last_error(void) { return g_err.c_str(); }
const char* skf_version(void) { return "skfusion_hip 0.1 (gfx950)"; }

int skf_plan_create(int32_t n_types, const skf_type_desc* types, int32_t n_relations,
                    const skf_relation_desc* relations, int32_t n_thetas, const skf_theta_desc* thetas,
                    const skf_options* opt, skf_plan** out) {
    return guarded([&] {
        if (!out || !types || !opt || n_types <= 0) SKF_FAIL(SKF_E_INVALID, "null argument / no object types");
        if (n_relations < 0 || n_thetas < 0 || (n_relations > 0 && !relations) || (n_thetas > 0 && !thetas))
            SKF_FAIL(SKF_E_INVALID, "bad relation / constraint arrays");
        if (opt->dtype != SKF_F64 && opt->dtype != SKF_F32 && opt->dtype != SKF_BF16)
            SKF_FAIL(SKF_E_INVALID, "unknown dtype %d", opt->dtype);
        if (opt->dtype == SKF_BF16 && opt->engine != SKF_ENGINE_MFMA)
            SKF_FAIL(SKF_E_INVALID, "SKF_BF16 needs the MFMA engine");
        if (opt->variant < SKF_DFMF || opt->variant > SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "bad variant");
        if (opt->engine != SKF_ENGINE_MFMA && opt->engine != SKF_ENGINE_VALU) SKF_FAIL(SKF_E_INVALID, "bad engine");
        skf_plan* p = new skf_plan();
        struct Guard { skf_plan* p; ~Guard() { delete p; } } guard{p};
        p->dtype = opt->dtype;
        p->variant = opt->variant;
        p->engine = opt->engine;
        p->f64 = (opt->dtype == SKF_F64);
        p->bf16 = (opt->dtype == SKF_BF16);
        p->esz = p->f64 ? 8 : 4;
        p->mt = p->f64 ? SKF_F64 : SKF_F32;
        p->target = opt->target_type;
        if (p->variant == SKF_TRANSFORM && (p->target < 0 || p->target >= n_types))
            SKF_FAIL(SKF_E_INVALID, "target type %d out of range", p->target);
        p->types.resize(n_types);
        for (int i = 0; i < n_types; ++i) {
            if (types[i].n_obj <= 0 || types[i].rank <= 0 || types[i].rank > EIGH_MAXN - 1)
                SKF_FAIL(SKF_E_INVALID, "object type %d: n_obj=%lld rank=%d invalid", i, (long long)types[i].n_obj,
                         types[i].rank);
            if (types[i].n_obj > 2000000000LL) SKF_FAIL(SKF_E_INVALID, "object type %d too large", i);
            p->types[i].n = types[i].n_obj;
            p->types[i].c = types[i].rank;
            p->types[i].n_pad = (types[i].rank + 1) / 2 * 2;
            p->types[i].t0 = 0;
            p->types[i].tn = types[i].n_obj;
            p->types[i].n_alloc = types[i].n_obj;
            if (opt->flags & SKF_OPT_OWNED_ROWS) {      // the rows this process owns (skf_owned_rows)
                if (opt->part_count < 1 || opt->part_index < 0 || opt->part_index >= opt->part_count)
                    SKF_FAIL(SKF_E_INVALID, "SKF_OPT_OWNED_ROWS: part_index %d outside [0, %d)", opt->part_index, opt->part_count);
                const int64_t ch = owned_chunk(opt->dtype, types[i].n_obj, opt->part_count);
                int64_t lo = ch * opt->part_index, hi = lo + ch;
                if (lo > types[i].n_obj) lo = types[i].n_obj;
                if (hi > types[i].n_obj) hi = types[i].n_obj;
                p->types[i].t0 = lo;
                p->types[i].tn = hi - lo;
                p->types[i].chunk = ch;
                p->types[i].n_alloc = ch * opt->part_count;
            } else if (opt->part_count > 1) {      // even shares of the rows, boundaries at multiples of 64
                if (opt->part_index < 0 || opt->part_index >= opt->part_count)
                    SKF_FAIL(SKF_E_INVALID, "part_index %d outside [0, %d)", opt->part_index, opt->part_count);
                const int64_t per = (types[i].n_obj + opt->part_count - 1) / opt->part_count;
                const int64_t step = (per + 63) / 64 * 64;
                int64_t lo = step * opt->part_index, hi = lo + step;
                if (lo > types[i].n_obj) lo = types[i].n_obj;
                if (hi > types[i].n_obj) hi = types[i].n_obj;
                p->types[i].t0 = lo;
                p->types[i].tn = hi - lo;
            }
        }
        if (opt->part_count > 1) p->sliced = true;
        if (opt->flags & SKF_OPT_OWNED_ROWS) {
            if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "SKF_OPT_OWNED_ROWS is for SKF_DFMF / SKF_DFMC plans");
            p->owned = p->sliced = true;
            p->part_index = opt->part_index;
            p->part_count = opt->part_count;
        }
        p->rels.resize(n_relations);
        for (int r = 0; r < n_relations; ++r) {
            const skf_relation_desc& d = relations[r];
            if (d.row_type < 0 || d.row_type >= n_types || d.col_type < 0 || d.col_type >= n_types)
                SKF_FAIL(SKF_E_INVALID, "relation %d: type index out of range", r);
            if (d.row_type == d.col_type) SKF_FAIL(SKF_E_INVALID, "relation %d: row type == column type (pass it as a constraint)", r);
            const bool absent = (d.flags & SKF_REL_ABSENT) != 0;
            if (!absent && (!d.data || d.ld < p->types[d.col_type].n))
                SKF_FAIL(SKF_E_INVALID, "relation %d: dimension mismatch (ld %lld < %lld columns)", r, (long long)d.ld,
                         (long long)p->types[d.col_type].n);
            const int64_t n_row_type = p->types[d.row_type].n;
            if (d.row_begin < 0 || d.n_rows < 0 || d.row_begin + d.n_rows > n_row_type)
                SKF_FAIL(SKF_E_INVALID, "relation %d: row block [%lld, +%lld) outside the %lld objects of its row type",
                         r, (long long)d.row_begin, (long long)d.n_rows, (long long)n_row_type);
            const bool block = absent || (d.n_rows > 0 && d.n_rows < n_row_type) || (d.flags & SKF_REL_NO_COL_SIDE);
            if (block && p->variant == SKF_TRANSFORM)
                SKF_FAIL(SKF_E_INVALID, "relation %d: row blocks are for SKF_DFMF / SKF_DFMC plans", r);
            if (block && p->bf16 && d.row_begin % 64 != 0)
                SKF_FAIL(SKF_E_INVALID, "relation %d: SKF_BF16 row blocks must start at a multiple of 64", r);
            if (block) p->sliced = true;
            if ((d.mask || (d.flags & SKF_REL_MASKED)) && p->variant != SKF_DFMC)
                SKF_FAIL(SKF_E_INVALID, "relation %d: masks need SKF_DFMC", r);
            if (d.mask && d.mask_ld < ((d.flags & SKF_REL_MASK_BITS) ? (p->types[d.col_type].n + 7) / 8 : p->types[d.col_type].n))
                SKF_FAIL(SKF_E_INVALID, "relation %d: mask ld", r);
            if (p->variant == SKF_TRANSFORM && d.row_type != p->target && d.col_type != p->target)
                SKF_FAIL(SKF_E_INVALID, "relation %d must include the target object type", r);
            RelState& s = p->rels[r];
            s.row = d.row_type; s.col = d.col_type;
            s.R_in = d.data; s.ld_in = d.ld; s.mask = d.mask; s.ldmask = d.mask_ld;
            s.mask_is_bits = (d.flags & SKF_REL_MASK_BITS) != 0;
            s.binary = p->bf16 && (d.flags & SKF_REL_BINARY) != 0 && !d.mask && !absent;
            s.R = d.data; s.ldr = d.ld;
            s.absent = absent;
            s.r0 = absent ? 0 : d.row_begin;
            s.nr = absent ? 0 : (d.n_rows > 0 ? d.n_rows : n_row_type - d.row_begin);
            s.col_side = (d.flags & SKF_REL_NO_COL_SIDE) == 0;
            s.masked = d.mask != nullptr || (d.flags & SKF_REL_MASKED) != 0;
            if (absent) { s.R_in = s.R = nullptr; s.mask = nullptr; }
            if (p->owned) {         // the block of a relation is the owned range of its row type, nothing else
                const TypeState& ti = p->types[d.row_type];
                if (ti.tn == 0 ? !absent : (absent || s.r0 != ti.t0 || s.nr != ti.tn))
                    SKF_FAIL(SKF_E_INVALID, "relation %d: SKF_OPT_OWNED_ROWS wants the rows [%lld, +%lld) of its row type here (skf_owned_rows)",
                             r, (long long)ti.t0, (long long)ti.tn);
                s.col_side = true;  // every process adds the column-side terms of ITS rows of the column type
            }
            if (d.known_bound < 0) SKF_FAIL(SKF_E_INVALID, "relation %d: negative bound on the known entries", r);
            s.kn_cap = (s.mask && p->variant == SKF_DFMC) ? d.known_bound : 0;
        }
        // masked relations with few known entries are kept as lists of those entries (skf_known.h).  The three passes over the
        // lists gather rank_row-wide vectors -- ~70 ps per entry at rank 128 against ~2.2 ps per CELL for the four passes of
        // the dense path over the completed relation (config 5, bf16) -- hence: known share * rank_row <= 4 (1/32 at rank
        // 128).  SKF_DFMC_SPARSE=0: never; =1: whenever a bound is given (up to a quarter of the relation).  Plans with row
        // blocks keep the dense form.
        {
            const Switches sw0 = Switches::read();          // (plan creation: the plan's own copy is read when its workspace is bound)
            const int mode = sw0.dfmc_sparse;
            // Parts of the lists (skf_known.h): with srp_bf16_v6_kernel the passes are no longer bound by instruction issue and
            // pinning slices of the gathered matrix to XCDs pays (profiles/r03_srp_v6.txt: 25.6 MB of user factors, 8 parts:
            // 1.75 -> 1.10 ms; 10 MB, 4 parts: 1.41 -> 1.25 ms) -- the smallest power of two that brings a slice under the
            // 4 MiB L2 of an XCD, as long as a segment still holds a batch of entries.  Other engines / widths: 1 (their
            // kernels are issue-bound; measured neutral in round 3).  SKF_KNOWN_PARTS=1|2|4|8 overrides.
            const int parts_env = sw0.known_parts;
            auto pick_parts = [&](int64_t n_in, int64_t n_out, int ci, int64_t cap) {
                if (parts_env) return parts_env;
                if (!p->bf16 || (ci != 128 && ci != 256)) return 1;
                int q = 1;
                while (q < 8 && (double)n_in * ci * 2.0 / q > 3.5 * 1048576.0) q *= 2;
                while (q > 1 && (double)cap / ((double)n_out * q) < 64.0) q /= 2;
                return q;
            };
            for (size_t rk = 0; rk < p->rels.size(); ++rk) {
                RelState& s = p->rels[rk];
                const int ci = p->types[s.row].c;
                if (p->owned) {
                    // ownership-aligned row blocks: the caller decided for ALL processes alike (SKF_REL_KNOWN_LISTS); a process
                    // without rows of the relation keeps the flag -- it adds the dense part of Q on ITS rows of the column type
                    const bool lists = (relations[rk].flags & SKF_REL_KNOWN_LISTS) != 0 && s.masked && p->variant == SKF_DFMC;
                    if (lists && (ci > 64 * SRP_MAXREP || s.kn_cap > 2000000000LL || (!s.absent && s.kn_cap <= 0)))
                        SKF_FAIL(SKF_E_INVALID, "relation %zu: SKF_REL_KNOWN_LISTS needs a bound on the known entries of the local rows", rk);
                    if (!lists) {
                        s.kn_cap = 0;
                        continue;
                    }
                } else {
                    if (s.kn_cap <= 0) continue;
                    const double cells = (double)s.nr * (double)p->types[s.col].n;
                    const double share = cells > 0 ? (double)s.kn_cap / cells : 1.0;
                    if (p->sliced || mode == 0 || share > 0.25 || (mode != 1 && share * ci > 4.0) || ci > 64 * SRP_MAXREP ||
                        s.kn_cap > 2000000000LL) {
                        s.kn_cap = 0;
                        continue;
                    }
                }
                s.kn = true;
                if (s.absent) continue;             // (no lists here: only the flag)
                s.kn_pc = pick_parts(p->types[s.col].n, s.nr, ci, s.kn_cap);        // row lists gather the column objects' vectors
                s.kn_pr = pick_parts(s.nr, p->types[s.col].n, ci, s.kn_cap);        // column lists gather the row objects' vectors
                s.kn_pw = ((p->types[s.col].n + s.kn_pc - 1) / s.kn_pc + 63) / 64 * 64;
                s.kn_ph = ((s.nr + s.kn_pr - 1) / s.kn_pr + 63) / 64 * 64;
                p->types[s.row].keep_prev = p->types[s.col].keep_prev = true;
                if (p->bf16) p->types[s.row].need_rows = true;              // the column lists gather the row type's bf16 rows
            }
            // sparse 0/1 relations as lists over bf16 factor rows (srp_bf16_v6_kernel<.., SRP_ONES>): both ranks 64 / 128 / 256
            auto gather_rank = [](int c) { return c == 64 || c == 128 || c == 256; };
            for (RelState& s : p->rels) {
                s.sp_gather = p->bf16 && s.binary && !s.masked && !s.absent && gather_rank(p->types[s.row].c) &&
                              gather_rank(p->types[s.col].c);
                if (!s.sp_gather) continue;
                p->types[s.row].need_rows = p->types[s.col].need_rows = true;
                // parts by the size of the gathered matrix alone (the number of ones is known at bind time, which may lower them)
                auto by_bytes = [&](int64_t n_in, int c) {
                    if (parts_env) return parts_env;
                    int q = 1;
                    while (q < 8 && (double)n_in * c * 2.0 / q > 3.5 * 1048576.0) q *= 2;
                    return q;
                };
                s.sp_pc = by_bytes(p->types[s.col].n, p->types[s.col].c);    // P: rows of G_j by the column index
                s.sp_pr = by_bytes(p->types[s.row].n, p->types[s.row].c);    // Q: rows of G_i by the row index
            }
        }
        p->thetas.resize(n_thetas);
        for (int t = 0; t < n_thetas; ++t) {
            if (thetas[t].type < 0 || thetas[t].type >= n_types || !thetas[t].data ||
                thetas[t].ld < p->types[thetas[t].type].n)
                SKF_FAIL(SKF_E_INVALID, "constraint %d invalid", t);
            if (p->variant == SKF_TRANSFORM && thetas[t].type != p->target)
                SKF_FAIL(SKF_E_INVALID, "constraint %d must be on the target object type", t);
            p->thetas[t].type = thetas[t].type;
            p->thetas[t].data = thetas[t].data;
            p->thetas[t].ld = thetas[t].ld;
            const int64_t nn = p->types[thetas[t].type].n;
            if (thetas[t].nnz < 0) SKF_FAIL(SKF_E_INVALID, "constraint %d: negative non-zero bound", t);
            if (thetas[t].nnz > 0 && thetas[t].nnz <= nn * nn / SKF_THETA_SPARSE_DIV) {
                p->thetas[t].sparse = true;
                p->thetas[t].nnz_cap = thetas[t].nnz;
            }
        }
        if (p->owned) {
            // SKF_BF16: the other owners' rows of a factor are read as bf16 operands only -- unless a constraint on the type
            // multiplies the f32 rows (theta_spmm_kernel / dense Theta).  Every type keeps its bf16 rows (the gathered form).
            for (TypeState& t : p->types) {
                t.gather_master = !p->bf16;
                if (p->bf16) t.need_rows = true;
            }
            for (const ThetaState& th : p->thetas) p->types[th.type].gather_master = true;
            // ... and the types of a masked relation (the same on every process, whatever it holds of the relation): the
            // known-entry form multiplies the f32 rows of both factors (cross-Gram matrices, T = G_j S^T, the dense part of Q)
            for (const RelState& r : p->rels)
                if (r.masked) p->types[r.row].gather_master = p->types[r.col].gather_master = true;
        }
        // ---- workspace layout: n-sized buffers in the master type, every c x c matrix in f64
        const size_t es = p->esz;
        size_t part_bytes = 0;
        size_t sp_part_bytes = 0;
        auto want_part = [&](int M, int N, int K, bool out_f64) {
            TileCfg t = pick_tile(out_f64, p->engine, M, N);
            int sl = pick_splits(t, M, N, K);
            if (!p->bf16 && (int64_t)cdiv(M, t.bm) * cdiv(N, t.bn) >= 256) {      // (relation contractions of the f32 / f64 engines)
                const int rs = pick_splits_relation(t, M, N, K, out_f64);
                if (rs > sl) sl = rs;
            }
            size_t need = (size_t)sl * (size_t)M * (size_t)N * (out_f64 ? 8 : 4);
            if (need > part_bytes) part_bytes = need;
        };
        int maxn = 2;
        // the E / D accumulators of all types form ONE contiguous range of the workspace, so that
        // a relation-sharded run sums them over the ranks with a single all-reduce
        // Three regions with the SAME internal layout -- all E, all D, all G (every type at the same offset in each) -- so
        // that the distributed iteration can cut them into `world` equal element ranges: reduce-scatter of E and of D,
        // update of the owned range of G, all-gather of G (skf_iterate_dist).  A pad behind each region takes the rounding
        // of the last range.
        p->acc_off = p->ws_bytes;
        for (int region = 0; region < 3; ++region) {
            const size_t begin = p->ws_bytes;
            for (int i = 0; i < n_types; ++i) {
                TypeState& t = p->types[i];
                const bool active = (p->variant != SKF_TRANSFORM) || i == p->target;
                if (region < 2 && !active) continue;
                add_slot(p, region == 0 ? t.E : region == 1 ? t.D : t.G, (size_t)t.n_alloc * t.c * es);
            }
            const size_t bytes = p->ws_bytes - begin;
            if (region == 0) { p->flat_e_off = begin; p->flat_bytes = bytes; }
            if (region == 1) p->flat_d_off = begin;
            if (region == 2) p->flat_g_off = begin;
            add_slot(p, p->flat_pad[region], 64 * 1024);
            if (region == 1) p->acc_bytes = p->ws_bytes - p->acc_off;
        }
        // the Gram matrices of all types form one range (SKF_OPT_OWNED_ROWS: one all-reduce of the partial sums)
        p->xg_off = p->ws_bytes;
        for (int i = 0; i < n_types; ++i) add_slot(p, p->types[i].Gram, (size_t)p->types[i].c * p->types[i].c * 8);
        p->xg_bytes = p->ws_bytes - p->xg_off;
        for (int i = 0; i < n_types; ++i) {
            TypeState& t = p->types[i];
            const bool active = (p->variant != SKF_TRANSFORM) || i == p->target;
            (void)active;
            if (p->bf16) {
                t.ldgt = pad64(t.n);
                add_slot(p, t.GTb, (size_t)t.c * t.ldgt * 2);
            }
            want_part(t.c, t.c, (int)t.n, true);
            if (t.keep_prev) add_slot(p, t.Gp, (size_t)t.n * t.c * es);
            if (p->bf16 && t.need_rows) {
                t.ldrow = (t.c + 7) / 8 * 8;
                add_slot(p, t.Grow, ((size_t)t.n_alloc + 1) * t.ldrow * 2);     // (+ 1: the all-zero row of the v6 list kernel)
            }
            if (p->variant != SKF_TRANSFORM) {
                add_slot(p, t.K, (size_t)t.c * t.c * 8);
                if (!p->f64) {
                    add_slot(p, t.Bp32, (size_t)t.c * t.c * 4);
                    add_slot(p, t.Bn32, (size_t)t.c * t.c * 4);
                }
                if (t.n_pad > maxn) maxn = t.n_pad;
            } else if (i == p->target) {
                add_slot(p, t.Ec, (size_t)t.n * t.c * es);
                add_slot(p, t.Dc, (size_t)t.n * t.c * es);
                add_slot(p, t.Galt, (size_t)t.n * t.c * es);
                add_slot(p, t.Bp_tot, (size_t)t.c * t.c * 8);
                add_slot(p, t.Bn_tot, (size_t)t.c * t.c * 8);
            }
        }
        size_t sq_elems = 1;
        if (p->variant != SKF_TRANSFORM) {
            // the per-type sums of B+ / B- form one range: one memset per iteration clears them all
            p->btot_off = p->ws_bytes;
            for (TypeState& t : p->types) {
                add_slot(p, t.Bp_tot, (size_t)t.c * t.c * 8);
                add_slot(p, t.Bn_tot, (size_t)t.c * t.c * 8);
            }
            p->btot_bytes = p->ws_bytes - p->btot_off;
            // exchange ranges of row-block sharding: all W; then Q of unmasked, then of masked relations
            p->xw_off = p->ws_bytes;
            for (RelState& r : p->rels) add_slot(p, r.W, (size_t)p->types[r.row].c * p->types[r.col].c * 8);
            p->xw_bytes = p->ws_bytes - p->xw_off;
            p->xq_off = p->ws_bytes;
            for (RelState& r : p->rels)
                if (!r.masked) add_slot(p, r.Q, (size_t)p->types[r.col].n_alloc * p->types[r.row].c * es);
            p->xq_bytes = p->ws_bytes - p->xq_off;
            p->xqm_off = p->ws_bytes;
            for (RelState& r : p->rels)
                if (r.masked) add_slot(p, r.Q, (size_t)p->types[r.col].n_alloc * p->types[r.row].c * es);
            p->xqm_bytes = p->ws_bytes - p->xqm_off;
        }
        for (RelState& r : p->rels) {
            TypeState& ti = p->types[r.row];
            TypeState& tj = p->types[r.col];
            const size_t cc = (size_t)ti.c * tj.c * 8;
            const int64_t nr = p->variant == SKF_TRANSFORM ? ti.n : r.nr;      // local rows
            if (p->variant == SKF_TRANSFORM) { r.nr = ti.n; r.r0 = 0; }
            add_slot(p, r.S, cc);
            add_slot(p, r.U, cc);
            if (nr > 0 && !r.kn) add_slot(p, r.H, (size_t)nr * tj.c * es);
            if (nr > 0 && !r.kn && (p->variant != SKF_TRANSFORM || r.row == p->target)) add_slot(p, r.P, (size_t)nr * tj.c * es);
            if (p->variant == SKF_TRANSFORM && r.col == p->target) add_slot(p, r.Q, (size_t)tj.n * ti.c * es);
            if (p->variant != SKF_TRANSFORM) {
                add_slot(p, r.T1, cc);
                if (!p->f64) add_slot(p, r.S32, cc / 2);
                if (nr > 0) want_part(ti.c, tj.c, (int)nr, true);
            }
            want_part(ti.c, tj.c, ti.c > tj.c ? ti.c : tj.c, true);
            want_part(ti.c, ti.c, tj.c, true);
            want_part(tj.c, tj.c, ti.c, true);
            if (nr <= 0 && r.kn) add_slot(p, r.U2, cc);             // (owned rows, none of this relation's here: the dense part of Q
                                                                    //  is still added on this process's rows of the column type)
            if (nr <= 0) continue;
            if (r.kn) {
                // the known entries as row lists and column lists, the gathered vectors, the row-side product, c x c scratch
                const size_t cap = (size_t)r.kn_cap;
                r.ldmb = (tj.n + 127) / 128 * 16;
                add_slot(p, r.Mb, (size_t)nr * r.ldmb);                         // (bind time only)
                add_slot(p, r.KrPtr, ((size_t)nr * r.kn_pc + 1) * 8);
                add_slot(p, r.KrIdx, cap * 4);
                add_slot(p, r.KrVal, cap * es);
                add_slot(p, r.KcPtr, ((size_t)tj.n * r.kn_pr + 1) * 8);
                add_slot(p, r.KcIdx, cap * 4);
                add_slot(p, r.KcVal, cap * es);
                add_slot(p, r.KcE, cap * es);
                const size_t cnt = std::max((size_t)nr * r.kn_pc, 2 * (size_t)tj.n * r.kn_pr);
                add_slot(p, r.KCnt, cnt * 4);
                r.kn_ldf = p->bf16 ? (ti.c + 7) / 8 * 8 : ti.c;
                if (p->bf16) add_slot(p, r.FiB, ((size_t)tj.n + 1) * r.kn_ldf * 2);  // (+ 1: the all-zero row of the v6 list kernel;
                                                                                    //  the row type's vectors: ti.Grow)
                add_slot(p, r.Tm, (size_t)tj.n * ti.c * es);
                add_slot(p, r.A, (size_t)nr * ti.c * es);
                if (r.kn_pc > 1) add_slot(p, r.Apart, (size_t)r.kn_pc * nr * ti.c * es);
                if (r.kn_pr > 1) add_slot(p, r.Qpart, (size_t)r.kn_pr * tj.n * ti.c * es);
                add_slot(p, r.Sp, cc);
                add_slot(p, r.U2, cc);
                add_slot(p, r.Xi, (size_t)ti.c * ti.c * 8);
                add_slot(p, r.Xj, (size_t)tj.c * tj.c * 8);
                add_slot(p, r.Bf, (size_t)ti.c * ti.c * 8);
                want_part(ti.c, tj.c, (int)tj.n, true);                         // W = Y^T G_j
                want_part(ti.c, ti.c, (int)nr, true);                           // G_i'^T G_i
                want_part(tj.c, tj.c, (int)tj.n, true);                         // G_j^T G_j'
                want_part((int)tj.n, ti.c, tj.c, p->f64);                       // T = G_j S^T ; Q += G_j (S^T Gram_i)
                want_part((int)nr, ti.c, ti.c, p->f64);                         // A += G_i (S Gram_j S^T)
                const size_t waves = (size_t)r.kn_pr * ((size_t)tj.n + 32) + 64;   // error partials: one per wave of the column pass
                if (waves > sq_elems) sq_elems = waves;
                continue;
            }
            if (r.mask && !p->bf16) add_slot(p, r.Rw, (size_t)nr * tj.n * es);
            if (r.mask) {
                r.ldmb = (tj.n + 127) / 128 * 16;               // bytes per packed mask row: whole 128-column tiles
                add_slot(p, r.Mb, (size_t)nr * r.ldmb);
            }
            if (p->bf16 && r.mask) {
                // known entries as compact per-tile lists, up to 1/8 of the relation (beyond that the completion
                // blends through the mask): 4 bytes per entry = at most a quarter of the bf16 relation's bytes
                const size_t tiles = (size_t)cdiv(nr, 256) * cdiv(tj.n, 128);     // (128-column tiles: the finer grid)
                r.kcap = (size_t)nr * tj.n / 8 + 4096;
                add_slot(p, r.Kcnt, tiles * 4);
                add_slot(p, r.Koff, (tiles + 1) * 4);
                add_slot(p, r.Klist, r.kcap * 4);
            }
            if (p->bf16) {                                       // completion / residual tiles
                r.ldhb = pad64(tj.c);
                add_slot(p, r.Hb, (size_t)nr * r.ldhb * 2);
                add_slot(p, r.Gb, (size_t)tj.n * r.ldhb * 2);
            }
            if (p->bf16) {
                r.ldrb = pad64(tj.n);
                r.kq = pad64(nr);
                if (r.binary) {
                    r.ldbb = r.ldrb / 8;
                    add_slot(p, r.Bb, (size_t)r.kq * r.ldbb);
                    if (!r.masked) {                       // room for the CSR / CSC form of a sparse relation
                        const int64_t per = r.sp_gather ? 80 : 256;            // (one entry in 80 / in 256 set, see build_sparse_pattern)
                        r.sp_cap = (int64_t)nr * tj.n / per > 0 ? (int64_t)nr * tj.n / per : 1;
                        if (r.sp_gather) {
                            if (r.sp_pc > 1) add_slot(p, r.SpRpP, ((size_t)nr * r.sp_pc + 1) * 8);
                            if (r.sp_pr > 1) add_slot(p, r.SpCpP, ((size_t)tj.n * r.sp_pr + 1) * 8);
                            const size_t pb = std::max(r.sp_pc > 1 ? (size_t)r.sp_pc * nr * tj.c * 4 : (size_t)0,
                                                       r.sp_pr > 1 ? (size_t)r.sp_pr * tj.n * ti.c * 4 : (size_t)0);
                            if (pb > sp_part_bytes) sp_part_bytes = pb;
                        }
                        add_slot(p, r.SpRp, (size_t)(nr + 1) * 8);
                        add_slot(p, r.SpCp, (size_t)(tj.n + 1) * 8);
                        add_slot(p, r.SpCi, (size_t)r.sp_cap * 4);
                        add_slot(p, r.SpRi, (size_t)r.sp_cap * 4);
                        add_slot(p, r.SpCnt, (size_t)(nr + 2 * tj.n + 8) * 4);
                    }
                } else {
                    add_slot(p, r.Rb, (size_t)r.kq * r.ldrb * 2);
                }
                size_t b1 = bf16_part_bytes((int)nr, tj.c, (int)r.ldrb, r.binary), b2 = bf16_part_bytes((int)tj.n, ti.c, (int)r.kq, true);
                if (b1 > part_bytes) part_bytes = b1;
                if (b2 > part_bytes) part_bytes = b2;
            }
            want_part((int)nr, tj.c, (int)tj.n, p->f64);
            want_part((int)tj.n, ti.c, (int)nr, p->f64);
            want_part((int)nr, ti.c, tj.c, p->f64);
            want_part((int)tj.n, tj.c, ti.c, p->f64);
            size_t blocks = (size_t)cdiv(nr, 32) * cdiv(tj.n, 32);           // smallest tile any engine uses
            if (blocks > sq_elems) sq_elems = blocks;
        }
        // small graphs: every rank <= 64, sparse constraints only, a few thousand objects per type -> the fused schedule
        {
            bool ok = p->variant == SKF_DFMF && p->engine == SKF_ENGINE_MFMA && !p->bf16 && !p->sliced && n_types <= SM_MAXT &&
                      n_relations >= 1 && n_relations <= SM_MAXR && n_thetas <= SM_MAXTH;
            // (object counts: the Q shares of the schedule grow with n_i / 256 * n_j * c_i, and from a few thousand objects
            // on the relation contractions are worth the big tiles of the general schedule)
            for (const TypeState& t : p->types) ok = ok && t.c <= SMALLC && t.n <= SM_MAX_OBJECTS;
            for (const ThetaState& th : p->thetas) ok = ok && th.sparse;
            for (const RelState& r : p->rels) ok = ok && !r.absent && !r.masked;
            p->small_fused = ok;
            if (ok) {
                size_t wdoubles = 0, gdoubles = 0;
                for (size_t k = 0; k < p->rels.size(); ++k) {
                    RelState& r = p->rels[k];
                    const TypeState& ti = p->types[r.row];
                    const TypeState& tj = p->types[r.col];
                    int part = 0;
                    for (int64_t r0 = 0; r0 < ti.n; r0 += 64) p->sm_j1.push_back(SmJob{SMJ_P, (int)k, (int)r0, (int)std::min<int64_t>(64, ti.n - r0), part++, 0, 0, 0});
                    // Q = R^T G_i: a long inner dimension (the rows of the relation) over a small output -> shares of SM_QROWS rows
                    int qpart = 0;
                    for (int64_t k0 = 0; k0 < ti.n; k0 += SM_QROWS, ++qpart)
                        for (int64_t c0 = 0; c0 < tj.n; c0 += 64)
                            p->sm_j1.push_back(SmJob{SMJ_Q, (int)k, (int)c0, (int)std::min<int64_t>(64, tj.n - c0), qpart, (int)k0,
                                                     (int)std::min<int64_t>(SM_QROWS, ti.n - k0), 0});
                    r.sm_qparts = qpart;
                    wdoubles += align_up((size_t)part * ti.c * tj.c, 16);      // (shares of different relations / types never share a cache line)
                    add_slot(p, r.SmQ, (size_t)qpart * tj.n * ti.c * es);
                    add_slot(p, r.SmBp, (size_t)ti.c * ti.c * 8);
                    add_slot(p, r.SmBn, (size_t)ti.c * ti.c * 8);
                    add_slot(p, r.SmDp, (size_t)tj.c * tj.c * 8);
                    add_slot(p, r.SmDn, (size_t)tj.c * tj.c * 8);
                }
                for (size_t i = 0; i < p->types.size(); ++i) {
                    const TypeState& t = p->types[i];
                    int part = 0;
                    for (int64_t r0 = 0; r0 < t.n; r0 += SM_GROWS) p->sm_j1.push_back(SmJob{SMJ_GRAM, (int)i, (int)r0, (int)std::min<int64_t>(SM_GROWS, t.n - r0), part++, 0, 0, 0});
                    gdoubles += align_up((size_t)part * t.c * t.c, 16);
                    for (int64_t r0 = 0; r0 < t.n; r0 += 64) p->sm_j3.push_back(SmJob{0, (int)i, (int)r0, (int)std::min<int64_t>(64, t.n - r0), 0, 0, 0, 0});
                    bool constrained = false;
                    for (const ThetaState& th : p->thetas) constrained = constrained || th.type == (int)i;
                    if (constrained)      // one wave per row, four rows per workgroup
                        for (int64_t r0 = 0; r0 < t.n; r0 += 4) p->sm_j1.push_back(SmJob{SMJ_THETA, (int)i, (int)r0, (int)std::min<int64_t>(4, t.n - r0), 0, 0, 0, 0});
                }
                // Gram shares first (the inverse hangs off the last of them), then P (W hangs off the last of those), Q, constraints
                auto prio = [](const SmJob& j) { return j.kind == SMJ_GRAM ? 0 : (j.kind == SMJ_P ? 1 : (j.kind == SMJ_Q ? 2 : 3)); };
                std::stable_sort(p->sm_j1.begin(), p->sm_j1.end(), [&](const SmJob& a, const SmJob& b) { return prio(a) < prio(b); });
                add_slot(p, p->sm_tables, sizeof(SmTables));
                add_slot(p, p->sm_jobs1, p->sm_j1.size() * sizeof(SmJob));
                add_slot(p, p->sm_jobs3, p->sm_j3.size() * sizeof(SmJob));
                add_slot(p, p->sm_wpart, wdoubles * 8);
                add_slot(p, p->sm_gpart, gdoubles * 8);
                add_slot(p, p->sm_tickets, (p->types.size() + p->rels.size()) * sizeof(int));
                add_slot(p, p->sm_batch, SKF_MAX_BATCH * sizeof(void*));          // table of tables: [0] = this plan's
            }
        }
        size_t theta_tmp_bytes = 0;
        for (ThetaState& th : p->thetas) {
            TypeState& t = p->types[th.type];
            if (th.sparse) {
                add_slot(p, th.Cnt, (size_t)t.n * sizeof(int));
                add_slot(p, th.Rp, (size_t)(t.n + 1) * sizeof(int64_t));
                add_slot(p, th.Ci, (size_t)th.nnz_cap * sizeof(int));
                add_slot(p, th.Vv, (size_t)th.nnz_cap * es);
                continue;
            }
            want_part((int)t.n, t.c, (int)t.n, p->f64);
            if (p->bf16) {
                th.ldb = pad64(t.n);
                add_slot(p, th.Pb, (size_t)t.n * th.ldb * 2);
                add_slot(p, th.Nb, (size_t)t.n * th.ldb * 2);
                const size_t b = bf16_part_bytes((int)t.n, t.c, (int)th.ldb, false);
                if (b > part_bytes) part_bytes = b;
                if ((size_t)t.n * t.c * 4 > theta_tmp_bytes) theta_tmp_bytes = (size_t)t.n * t.c * 4;
            }
        }
        if (!p->thetas.empty()) add_slot(p, p->theta_flags, p->thetas.size() * 2 * sizeof(int));
        if (theta_tmp_bytes) add_slot(p, p->theta_tmp, theta_tmp_bytes);
        p->part_bytes = part_bytes;
        add_slot(p, p->part, part_bytes);
        if (sp_part_bytes) add_slot(p, p->sp_part, sp_part_bytes);
        size_t aux_bytes = 0;
        for (TypeState& t : p->types) {
            TileCfg tc = pick_tile(true, p->engine, t.c, t.c);
            size_t need = (size_t)pick_splits(tc, t.c, t.c, (int)t.n) * (size_t)t.c * t.c * 8;
            if (need > aux_bytes) aux_bytes = need;
        }
        for (RelState& r : p->rels) {          // W = G_i^T P on the second stream (pipelined schedule)
            const int ci = p->types[r.row].c, cj = p->types[r.col].c;
            TileCfg tc = pick_tile(true, p->engine, ci, cj);
            size_t need = (size_t)pick_splits(tc, ci, cj, (int)r.nr) * (size_t)ci * cj * 8;
            if (need > aux_bytes) aux_bytes = need;
        }
        p->part_aux_bytes = aux_bytes;
        add_slot(p, p->part_aux, aux_bytes);
        p->sq_elems = sq_elems;
        add_slot(p, p->sqpart, sq_elems * 8);
        if (p->variant != SKF_TRANSFORM) {
            p->eig_maxn = maxn;
            p->eig_stride = (int64_t)maxn * maxn;
            const size_t mat = (size_t)p->eig_stride * n_types * sizeof(double);
            add_slot(p, p->eigA, mat);
            add_slot(p, p->eigV, mat);
            add_slot(p, p->eigVs, mat);
            add_slot(p, p->eigW, (size_t)maxn * n_types * sizeof(double));
            add_slot(p, p->eigN, (size_t)n_types * sizeof(int));
            add_slot(p, p->eigNorig, (size_t)n_types * sizeof(int));
            add_slot(p, p->eigOk, (size_t)n_types * sizeof(int));
            if (maxn > SWEEP_MAXN && n_types <= PINV_MAXB) add_slot(p, p->eigX, defl_scratch_bytes(n_types, p->eig_stride));
        }
        guard.p = nullptr;
        *out = p;
    });
}

int skf_plan_destroy(skf_plan* plan) {
    delete plan;
    return SKF_OK;
}

int skf_plan_workspace_bytes(const skf_plan* plan, size_t* bytes) {
    return guarded([&] {
        if (!plan || !bytes) SKF_FAIL(SKF_E_INVALID, "null argument");
        *bytes = plan->ws_bytes;
    });
}

// CSR + CSC of a very sparse binary relation from its bitmap (bind time): per-row counts on the device, prefix sums on the
// host; kept only when the ones fit the slots sized at plan creation (1 entry in 256)
static void build_sparse_pattern(skf_plan* p, RelState& r, hipStream_t st) {
    r.sparse = false;
    if (r.sp_cap <= 0 || !r.SpRp.ptr) return;
    const int64_t rows = r.nr, cols = p->types[r.col].n;
    if (rows <= 0 || cols <= 0) return;
    const int wgrid = (int)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
    int* rowcnt = (int*)r.SpCnt.ptr;
    int* colcnt = rowcnt + rows;
    int* fillpos = colcnt + cols;
    hipLaunchKernelGGL(bits_row_count_kernel, dim3(wgrid), dim3(256), 0, st, (const uint8_t*)r.Bb.ptr, r.ldbb, rows, rowcnt);
    check_launch("bits_row_count");
    std::vector<int> cnt((size_t)(rows > cols ? rows : cols));
    SKF_HIP(hipMemcpyAsync(cnt.data(), rowcnt, (size_t)rows * 4, hipMemcpyDeviceToHost, st));
    SKF_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> ptr((size_t)(rows > cols ? rows : cols) + 1);
    int64_t tot = 0;
    for (int64_t k = 0; k < rows; ++k) { ptr[k] = tot; tot += cnt[k]; }
    ptr[rows] = tot;
    // the form by the count: lists over bf16 factor rows (srp_bf16_v6_kernel<.., SRP_ONES>) up to 1 entry in 80 -- measured
    // at config 5 (profiles/r03_srp_v6.txt): ~14-22 ps per one and contraction against ~0.28 ps per CELL of the bitmap
    // kernels, break-even near 1 in 64 --; without that form (other ranks) lists over the f32 rows up to 1 in 256
    if (tot > r.sp_cap || (!r.sp_gather && tot > rows * cols / 256)) return;
    r.sp_nnz = tot;
    SKF_HIP(hipMemcpyAsync(r.SpRp.ptr, ptr.data(), (size_t)(rows + 1) * 8, hipMemcpyHostToDevice, st));
    SKF_HIP(hipMemsetAsync(colcnt, 0, (size_t)cols * 2 * 4, st));
    SKF_HIP(hipStreamSynchronize(st));                       // (`ptr` is reused below)
    if (tot > 0) {
        hipLaunchKernelGGL(bits_csr_fill_kernel, dim3(wgrid), dim3(256), 0, st, (const uint8_t*)r.Bb.ptr, r.ldbb, rows,
                           (const int64_t*)r.SpRp.ptr, (int*)r.SpCi.ptr);
        hipLaunchKernelGGL(csr_col_count_kernel, dim3(elem_grid(tot)), dim3(256), 0, st, (const int*)r.SpCi.ptr, tot, colcnt);
        check_launch("bits_csr_fill");
    }
    SKF_HIP(hipMemcpyAsync(cnt.data(), colcnt, (size_t)cols * 4, hipMemcpyDeviceToHost, st));
    SKF_HIP(hipStreamSynchronize(st));
    int64_t t2 = 0;
    for (int64_t k = 0; k < cols; ++k) { ptr[k] = t2; t2 += cnt[k]; }
    ptr[cols] = t2;
    SKF_HIP(hipMemcpyAsync(r.SpCp.ptr, ptr.data(), (size_t)(cols + 1) * 8, hipMemcpyHostToDevice, st));
    if (tot > 0) {
        hipLaunchKernelGGL(csr_transpose_fill_kernel, dim3(wgrid), dim3(256), 0, st, (const int64_t*)r.SpRp.ptr,
                           (const int*)r.SpCi.ptr, rows, (const int64_t*)r.SpCp.ptr, fillpos, (int*)r.SpRi.ptr);
        hipLaunchKernelGGL(csc_sort_kernel, dim3(elem_grid(cols)), dim3(256), 0, st, (const int64_t*)r.SpCp.ptr,
                           (int*)r.SpRi.ptr, cols);
        check_launch("csc_build");
    }
    if (r.sp_gather) {          // lists in parts pinned to XCDs, as long as a segment still holds a batch of entries
        const bool forced = p->sw.known_parts_forced;                    // (tests: short lists in parts too)
        auto fit = [&](int q, int64_t n_out) {
            while (!forced && q > 1 && (double)tot / ((double)n_out * q) < 64.0) q /= 2;
            return q;
        };
        r.sp_pc = fit(r.SpRpP.ptr ? r.sp_pc : 1, rows);
        r.sp_pr = fit(r.SpCpP.ptr ? r.sp_pr : 1, cols);
        r.sp_pw = ((cols + r.sp_pc - 1) / r.sp_pc + 63) / 64 * 64;
        r.sp_ph = ((rows + r.sp_pr - 1) / r.sp_pr + 63) / 64 * 64;
        if (r.sp_pc > 1)
            hipLaunchKernelGGL(parted_ptr_kernel, dim3(elem_grid(rows * r.sp_pc + 1)), dim3(256), 0, st, (const int64_t*)r.SpRp.ptr,
                               (const int*)r.SpCi.ptr, rows, r.sp_pc, r.sp_pw, (int64_t*)r.SpRpP.ptr);
        if (r.sp_pr > 1)
            hipLaunchKernelGGL(parted_ptr_kernel, dim3(elem_grid(cols * r.sp_pr + 1)), dim3(256), 0, st, (const int64_t*)r.SpCp.ptr,
                               (const int*)r.SpRi.ptr, cols, r.sp_pr, r.sp_ph, (int64_t*)r.SpCpP.ptr);
        check_launch("parted_ptr");
    }
    SKF_HIP(hipStreamSynchronize(st));
    r.sparse = true;
}

int skf_plan_bind_workspace(skf_plan* p, void* ws, size_t bytes, void* stream) {
    return guarded([&] {
        if (!p || !ws) SKF_FAIL(SKF_E_INVALID, "null argument");
        if (bytes < p->ws_bytes) SKF_FAIL(SKF_E_WORKSPACE, "workspace %zu B < required %zu B", bytes, p->ws_bytes);
        if (((uintptr_t)ws & 255) != 0) SKF_FAIL(SKF_E_WORKSPACE, "workspace must be 256-byte aligned");
        for (Slot* s : p->slots) s->ptr = (char*)ws + s->off;
        p->ws_base = ws;
        p->sw = Switches::read();          // the only place a plan looks at the environment
        hipStream_t st = as_stream(stream);
        for (RelState& r : p->rels) {
            if (!r.mask) continue;
            // the mask in the engine's layout: one bit per entry, rows padded to whole 128-column tiles.
            // The caller's mask (bytes or bits) is not referenced after this call.
            const int64_t rows = r.nr, cols = p->types[r.col].n;
            if (r.mask_is_bits) {
                SKF_HIP(hipMemsetAsync(r.Mb.ptr, 0, r.Mb.bytes, st));
                hipLaunchKernelGGL(copy_mask_bits_kernel, dim3(elem_grid(rows * ((cols + 7) / 8))), dim3(256), 0, st,
                                   (uint8_t*)r.Mb.ptr, r.ldmb, r.mask, r.ldmask, rows, cols);
            } else {
                hipLaunchKernelGGL(pack_mask_kernel, dim3(elem_grid(rows * r.ldmb)), dim3(256), 0, st, (uint8_t*)r.Mb.ptr,
                                   r.ldmb, r.mask, r.ldmask, rows, cols);
            }
            check_launch("pack_mask");
            if (r.kn) {                            // the known entries as lists; no working copy of the relation
                build_known_lists(p, r, st);
                continue;
            }
            if (p->bf16) {
                // known entries of every tile of the completion pass (256 rows x epi_tile columns) as a compact list
                // (count, prefix sum on the host, fill)
                const int tx = cdiv(rows, 256), ty = cdiv(cols, 128);
                const size_t tiles = (size_t)tx * ty;
                KnownArgs ka;
                ka.mbits = (const uint8_t*)r.Mb.ptr; ka.ldmb = r.ldmb;
                ka.Rin = (const uint16_t*)r.R_in; ka.ldin = r.ld_in;
                ka.rows = (int)rows; ka.cols = (int)cols;
                ka.tile_cols = 128;
                ka.counts = (uint32_t*)r.Kcnt.ptr; ka.off = nullptr; ka.list = nullptr;
                hipLaunchKernelGGL(known_entries_kernel, dim3(tx, ty), dim3(256), 0, st, ka);
                check_launch("known_entries(count)");
                std::vector<uint32_t> cnt(tiles), off(tiles + 1);
                SKF_HIP(hipMemcpyAsync(cnt.data(), r.Kcnt.ptr, tiles * 4, hipMemcpyDeviceToHost, st));
                SKF_HIP(hipStreamSynchronize(st));
                uint64_t tot = 0;
                for (size_t t = 0; t < tiles; ++t) { off[t] = (uint32_t)tot; tot += cnt[t]; }
                off[tiles] = (uint32_t)tot;
                r.use_klist = tot <= r.kcap && tot < 0xFFFFFFFFull;
                if (r.use_klist) {
                    SKF_HIP(hipMemcpyAsync(r.Koff.ptr, off.data(), (tiles + 1) * 4, hipMemcpyHostToDevice, st));
                    ka.off = (const uint32_t*)r.Koff.ptr; ka.list = (uint32_t*)r.Klist.ptr;
                    hipLaunchKernelGGL(known_entries_kernel, dim3(tx, ty), dim3(256), 0, st, ka);
                    check_launch("known_entries(fill)");
                    SKF_HIP(hipStreamSynchronize(st));          // `off` dies here; bind is not on the hot path
                }
                continue;                          // bf16: the padded copy below is the working set
            }
            copy2d(r.Rw.ptr, cols, r.R_in, r.ld_in, rows, cols, p->esz, st);
            r.R = r.Rw.ptr;
            r.ldr = cols;
        }
        if (p->bf16) {
            // the caller's bf16 relation is copied ONCE into a zero-padded row-major layout (rows to a multiple of
            // 64: the inner dimension of Q = R^T G_i; columns to a multiple of 64: the inner dimension of
            // P = R G_j); it is not referenced after this call
            for (TypeState& t : p->types) {
                SKF_HIP(hipMemsetAsync(t.GTb.ptr, 0, t.GTb.bytes, st));
                if (t.Grow.bytes) SKF_HIP(hipMemsetAsync(t.Grow.ptr, 0, t.Grow.bytes, st));
            }
            for (RelState& r : p->rels) {
                if (r.absent || r.kn) continue;
                const int64_t rows = r.nr, cols = p->types[r.col].n;
                if (r.binary) {
                    int* bad = (int*)p->sqpart.ptr;                  // (scratch word; bind is not on the hot path)
                    SKF_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
                    hipLaunchKernelGGL(pack_binary_kernel, dim3(elem_grid(r.kq * r.ldbb)), dim3(256), 0, st, (uint8_t*)r.Bb.ptr,
                                       r.ldbb, r.kq, (const uint16_t*)r.R_in, r.ld_in, rows, cols, bad);
                    check_launch("pack_binary");
                    int hbad = 0;
                    SKF_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
                    SKF_HIP(hipStreamSynchronize(st));
                    if (hbad) SKF_FAIL(SKF_E_INVALID, "a relation flagged SKF_REL_BINARY holds an entry that is neither 0 nor 1");
                    if (r.Hb.bytes) {
                        SKF_HIP(hipMemsetAsync(r.Hb.ptr, 0, r.Hb.bytes, st));
                        SKF_HIP(hipMemsetAsync(r.Gb.ptr, 0, r.Gb.bytes, st));
                    }
                    r.R = r.Bb.ptr;
                    r.ldr = r.ldrb;
                    build_sparse_pattern(p, r, st);
                    continue;
                }
                SKF_HIP(hipMemsetAsync(r.Rb.ptr, 0, r.Rb.bytes, st));
                if (r.Hb.bytes) {
                    SKF_HIP(hipMemsetAsync(r.Hb.ptr, 0, r.Hb.bytes, st));
                    SKF_HIP(hipMemsetAsync(r.Gb.ptr, 0, r.Gb.bytes, st));
                }
                launch_to_bf16<uint16_t>((uint16_t*)r.Rb.ptr, r.ldrb, (const uint16_t*)r.R_in, r.ld_in, rows, cols, false, st);
                r.R = r.Rb.ptr;
                r.ldr = r.ldrb;
            }
        }
        for (ThetaState& th : p->thetas) {
            if (!th.sparse) continue;
            // CSR of a sparse constraint: per-row counts on the device, prefix sum on the host, fill on the device
            const int64_t n = p->types[th.type].n;
            const int grid = (int)((n + 3) / 4 < 2048 ? (n + 3) / 4 : 2048);
            if (p->f64)
                hipLaunchKernelGGL((theta_row_count_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)th.data, th.ld, n, (int*)th.Cnt.ptr);
            else
                hipLaunchKernelGGL((theta_row_count_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)th.data, th.ld, n, (int*)th.Cnt.ptr);
            check_launch("theta_row_count");
            std::vector<int> cnt((size_t)n);
            SKF_HIP(hipMemcpyAsync(cnt.data(), th.Cnt.ptr, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
            SKF_HIP(hipStreamSynchronize(st));
            std::vector<int64_t> rp((size_t)n + 1);
            int64_t tot = 0;
            for (int64_t r = 0; r < n; ++r) { rp[(size_t)r] = tot; tot += cnt[(size_t)r]; }
            rp[(size_t)n] = tot;
            if (tot > th.nnz_cap)
                SKF_FAIL(SKF_E_INVALID, "constraint on type %d holds %lld non-zeros, more than the bound %lld given in skf_theta_desc.nnz",
                         th.type, (long long)tot, (long long)th.nnz_cap);
            th.nnz = tot;
            SKF_HIP(hipMemcpyAsync(th.Rp.ptr, rp.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
            if (p->f64)
                hipLaunchKernelGGL((theta_csr_fill_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)th.data, th.ld, n,
                                   (const int64_t*)th.Rp.ptr, (int*)th.Ci.ptr, (double*)th.Vv.ptr);
            else
                hipLaunchKernelGGL((theta_csr_fill_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)th.data, th.ld, n,
                                   (const int64_t*)th.Rp.ptr, (int*)th.Ci.ptr, (float*)th.Vv.ptr);
            check_launch("theta_csr_fill");
            SKF_HIP(hipStreamSynchronize(st));          // `rp` dies here; bind is not on the hot path
        }
        if (!p->thetas.empty()) {
            // which halves of every constraint's +- split are non-empty (one device pass, read back here:
            // bind is not on the hot path), and the bf16 engine's copies of the non-empty halves
            SKF_HIP(hipMemsetAsync(p->theta_flags.ptr, 0, p->theta_flags.bytes, st));
            for (size_t k = 0; k < p->thetas.size(); ++k) {
                ThetaState& th = p->thetas[k];
                if (th.sparse) continue;
                const int64_t n = p->types[th.type].n;
                int* fl = (int*)p->theta_flags.ptr + 2 * k;
                if (p->f64)
                    hipLaunchKernelGGL((sign_flags_kernel<double>), dim3(elem_grid(n * n)), dim3(256), 0, st,
                                       (const double*)th.data, th.ld, n, n, fl);
                else
                    hipLaunchKernelGGL((sign_flags_kernel<float>), dim3(elem_grid(n * n)), dim3(256), 0, st,
                                       (const float*)th.data, th.ld, n, n, fl);
                check_launch("sign_flags");
            }
            std::vector<int> flags(p->thetas.size() * 2);
            SKF_HIP(hipMemcpyAsync(flags.data(), p->theta_flags.ptr, flags.size() * sizeof(int), hipMemcpyDeviceToHost, st));
            SKF_HIP(hipStreamSynchronize(st));
            for (size_t k = 0; k < p->thetas.size(); ++k) {
                ThetaState& th = p->thetas[k];
                if (th.sparse) continue;
                th.has_pos = flags[2 * k] != 0;
                th.has_neg = flags[2 * k + 1] != 0;
                if (!p->bf16) continue;
                const int64_t n = p->types[th.type].n;
                for (int half = 0; half < 2; ++half) {
                    if (!(half == 0 ? th.has_pos : th.has_neg)) continue;
                    Slot& dst = half == 0 ? th.Pb : th.Nb;
                    SKF_HIP(hipMemsetAsync(dst.ptr, 0, dst.bytes, st));
                    hipLaunchKernelGGL(split_to_bf16_kernel, dim3(elem_grid(n * n)), dim3(256), 0, st, (uint16_t*)dst.ptr,
                                       th.ldb, (const float*)th.data, th.ld, n, n, half == 0 ? AOP_POS : AOP_NEG);
                    check_launch("split_to_bf16");
                }
            }
        }
        if (p->variant != SKF_TRANSFORM) {
            std::vector<int> n_pad, n_orig;
            for (TypeState& t : p->types) {
                n_pad.push_back(t.n_pad);
                n_orig.push_back(t.c);
            }
            SKF_HIP(hipMemcpyAsync(p->eigN.ptr, n_pad.data(), n_pad.size() * sizeof(int), hipMemcpyHostToDevice, st));
            SKF_HIP(hipMemcpyAsync(p->eigNorig.ptr, n_orig.data(), n_orig.size() * sizeof(int), hipMemcpyHostToDevice, st));
            SKF_HIP(hipStreamSynchronize(st));     // the host vectors die here; bind is not on the hot path
        }
        if (p->small_fused && (p->sw.no_small_fused || p->sw.no_small_chain)) p->small_fused = false;
        if (p->small_fused) {
            // job tables and the pointer table of the fused small-graph schedule (skf_small.h)
            SmTables tb;
            memset(&tb, 0, sizeof tb);
            tb.n_types = (int)p->types.size(); tb.n_rels = (int)p->rels.size(); tb.n_thetas = (int)p->thetas.size();
            tb.nan_upd = 1;                                   // DFMF: nan_to_num on the A / B / C / D terms (_dfmf.py:254-276)
            tb.wpart = (double*)p->sm_wpart.ptr; tb.gpart = (double*)p->sm_gpart.ptr;
            tb.tickets = (int*)p->sm_tickets.ptr;
            SKF_HIP(hipMemsetAsync(p->sm_tickets.ptr, 0, p->sm_tickets.bytes, st));
            tb.eigA = (double*)p->eigA.ptr; tb.eigV = (double*)p->eigV.ptr; tb.eigOk = (int*)p->eigOk.ptr;
            tb.eig_stride = p->eig_stride;
            tb.chol_thr = chol_rel_threshold(p->sw);
            tb.eig.A = (double*)p->eigA.ptr; tb.eig.V = (double*)p->eigV.ptr; tb.eig.Vs = (double*)p->eigVs.ptr;
            tb.eig.w = (double*)p->eigW.ptr; tb.eig.stride = p->eig_stride; tb.eig.wstride = p->eig_maxn;
            tb.eig.n = (const int*)p->eigN.ptr; tb.eig.n_orig = (const int*)p->eigNorig.ptr;
            tb.eig.chol_ok = (int*)p->eigOk.ptr;
            tb.eig.max_sweeps = 30;
            tb.defl_lo = deflation_lo(p->sw); tb.defl_hi = 1e-7;
            tb.lds_rank = p->eig_maxn < 64 ? p->eig_maxn : 64;          // packed r (r + 1) / 2 doubles inside the staging tiles
            tb.sweep_single = p->sw.small_sweep1 ? 1 : 0;
            int64_t goff = 0, woff = 0;
            for (size_t i = 0; i < p->types.size(); ++i) {
                TypeState& t = p->types[i];
                SmType& d = tb.t[i];
                d.G = t.G.ptr; d.E = t.E.ptr; d.D = t.D.ptr; d.Gram = (double*)t.Gram.ptr; d.K = (double*)t.K.ptr;
                d.n = t.n; d.c = t.c; d.gpart_off = goff; d.n_gjobs = (int)((t.n + SM_GROWS - 1) / SM_GROWS);
                goff += (int64_t)align_up((size_t)d.n_gjobs * t.c * t.c, 16);
                for (const ThetaState& th : p->thetas) d.has_theta = d.has_theta || th.type == (int)i;
            }
            for (size_t k = 0; k < p->rels.size(); ++k) {
                RelState& r = p->rels[k];
                SmRel& d = tb.r[k];
                d.R = r.R; d.ldr = r.ldr; d.P = r.P.ptr; d.Q = r.SmQ.ptr; d.n_qparts = r.sm_qparts; d.W = (double*)r.W.ptr; d.S = (double*)r.S.ptr;
                d.Bp = (double*)r.SmBp.ptr; d.Bn = (double*)r.SmBn.ptr; d.Dp = (double*)r.SmDp.ptr; d.Dn = (double*)r.SmDn.ptr;
                d.row = r.row; d.col = r.col; d.wpart_off = woff; d.n_pjobs = (int)((p->types[r.row].n + 63) / 64);
                woff += (int64_t)align_up((size_t)d.n_pjobs * p->types[r.row].c * p->types[r.col].c, 16);
            }
            for (size_t k = 0; k < p->thetas.size(); ++k) {
                ThetaState& th = p->thetas[k];
                tb.th[k].rp = (const int64_t*)th.Rp.ptr; tb.th[k].ci = (const int*)th.Ci.ptr; tb.th[k].vv = th.Vv.ptr;
                tb.th[k].type = th.type;
            }
            SKF_HIP(hipMemcpyAsync(p->sm_tables.ptr, &tb, sizeof tb, hipMemcpyHostToDevice, st));
            p->sm_batch_host.assign(1, p->sm_tables.ptr);
            SKF_HIP(hipMemcpyAsync(p->sm_batch.ptr, p->sm_batch_host.data(), sizeof(void*), hipMemcpyHostToDevice, st));
            SKF_HIP(hipMemcpyAsync(p->sm_jobs1.ptr, p->sm_j1.data(), p->sm_j1.size() * sizeof(SmJob), hipMemcpyHostToDevice, st));
            SKF_HIP(hipMemcpyAsync(p->sm_jobs3.ptr, p->sm_j3.data(), p->sm_j3.size() * sizeof(SmJob), hipMemcpyHostToDevice, st));
            SKF_HIP(hipStreamSynchronize(st));     // (`tb` dies here; bind is not on the hot path)
        }
        if (p->owned) {
            // padded layouts of the exchanges: rows past the objects of a type stay zero for good (they are gathered and
            // scattered with the rest), the Gram range is summed as a whole
            for (TypeState& t : p->types) {
                SKF_HIP(hipMemsetAsync(t.G.ptr, 0, t.G.bytes, st));
                SKF_HIP(hipMemsetAsync(t.E.ptr, 0, t.E.bytes, st));
                SKF_HIP(hipMemsetAsync(t.D.ptr, 0, t.D.bytes, st));
            }
            for (RelState& r : p->rels) SKF_HIP(hipMemsetAsync(r.Q.ptr, 0, r.Q.bytes, st));
            SKF_HIP(hipMemsetAsync((char*)ws + p->xg_off, 0, p->xg_bytes, st));
            SKF_HIP(hipMemsetAsync((char*)ws + p->xw_off, 0, p->xw_bytes, st));
            if (!p->cs && !p->sw.no_overlap && p->sw.comm_stream)
                SKF_HIP(hipStreamCreateWithFlags(&p->cs, hipStreamNonBlocking));
        }
        p->pipeline = !p->sw.no_pipeline;
        // (a plan on the three-launch schedule of small graphs issues everything on the caller's stream: no second stream to
        // create and destroy -- at ten restarts of the README graph the streams of the plans were 4 of 30 ms)
        if (p->variant != SKF_TRANSFORM && !p->aux && !p->small_fused) {
            if (!p->sw.no_overlap) {
                {   // the second stream at the LOWEST priority: its launches fill what the contractions of the main stream
                    // leave free instead of taking CUs from them (config 5 +0.9 %, config 3 +0.5 % against the default priority)
                    int lo = 0, hi = 0;
                    SKF_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
                    // (plans with owned rows, where the second stream carries the critical path of a rank: lowest / default /
                    // highest priority measured equal -- 2.58 / 2.56 / 2.55 ms for rank 3 of 8 at config 3 --, a running
                    // contraction workgroup is not preempted; profiles/r04_owned_rank_emulation.txt)
                    SKF_HIP(hipStreamCreateWithPriority(&p->aux, hipStreamNonBlocking, lo));
                }
                SKF_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
                SKF_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
                p->overlap = true;
            }
        }
        if (p->graph_exec) {
            (void)hipGraphExecDestroy(p->graph_exec);
            p->graph_exec = nullptr;
        }
        p->graph_failed = false;
        p->bound = true;
        p->prepared = false;
        p->first_iter = true;
        p->kn_first = true;
    });
}

static void check_bound(const skf_plan* p) {
    if (!p) SKF_FAIL(SKF_E_INVALID, "null plan");
    if (!p->bound) SKF_FAIL(SKF_E_STATE, "workspace not bound");
}

int skf_set_factor(skf_plan* p, int32_t type, const void* G, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (type < 0 || type >= (int)p->types.size() || !G) SKF_FAIL(SKF_E_INVALID, "bad type index / pointer");
        TypeState& t = p->types[type];
        if (ld < t.c) SKF_FAIL(SKF_E_INVALID, "ld %lld < rank %d", (long long)ld, t.c);
        copy2d(t.G.ptr, t.c, G, ld, t.n, t.c, p->esz, as_stream(stream));
        refresh_gt(p, t, as_stream(stream));
        t.set = true;
        p->prepared = false;
    });
}

int skf_get_factor(const skf_plan* p, int32_t type, void* G, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (type < 0 || type >= (int)p->types.size() || !G) SKF_FAIL(SKF_E_INVALID, "bad type index / pointer");
        const TypeState& t = p->types[type];
        if (ld < t.c) SKF_FAIL(SKF_E_INVALID, "ld %lld < rank %d", (long long)ld, t.c);
        copy2d(G, ld, t.G.ptr, t.c, t.n, t.c, p->esz, as_stream(stream));
    });
}

int skf_set_backbone(skf_plan* p, int32_t rel, const void* S, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (rel < 0 || rel >= (int)p->rels.size() || !S) SKF_FAIL(SKF_E_INVALID, "bad relation index / pointer");
        RelState& r = p->rels[rel];
        const int ci = p->types[r.row].c, cj = p->types[r.col].c;
        if (ld < cj) SKF_FAIL(SKF_E_INVALID, "ld too small");
        if (p->f64) {
            copy2d(r.S.ptr, cj, S, ld, ci, cj, 8, as_stream(stream));
        } else {
            hipLaunchKernelGGL((cast_kernel<double, float>), dim3(elem_grid((int64_t)ci * cj)), dim3(256), 0,
                               as_stream(stream), (double*)r.S.ptr, (int64_t)cj, (const float*)S, ld, (int64_t)ci,
                               (int64_t)cj);
            check_launch("cast");
        }
        r.s_set = true;
        p->prepared = false;
    });
}

int skf_get_backbone(const skf_plan* p, int32_t rel, void* S, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (rel < 0 || rel >= (int)p->rels.size() || !S) SKF_FAIL(SKF_E_INVALID, "bad relation index / pointer");
        const RelState& r = p->rels[rel];
        const int ci = p->types[r.row].c, cj = p->types[r.col].c;
        if (ld < cj) SKF_FAIL(SKF_E_INVALID, "ld too small");
        if (p->f64) {
            copy2d(S, ld, r.S.ptr, cj, ci, cj, 8, as_stream(stream));
        } else {
            hipLaunchKernelGGL((cast_kernel<float, double>), dim3(elem_grid((int64_t)ci * cj)), dim3(256), 0,
                               as_stream(stream), (float*)S, ld, (const double*)r.S.ptr, (int64_t)cj, (int64_t)ci,
                               (int64_t)cj);
            check_launch("cast");
        }
    });
}

int skf_iterate(skf_plan* p, int32_t n_iters, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->sliced) SKF_FAIL(SKF_E_STATE, "a plan with row blocks iterates through skf_stage + the exchanges");
        if (n_iters < 0) SKF_FAIL(SKF_E_INVALID, "n_iters < 0");
        for (size_t i = 0; i < p->types.size(); ++i)
            if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "factor of object type %zu not set", i);
        hipStream_t st = as_stream(stream);
        if (p->variant == SKF_TRANSFORM) {
            for (size_t r = 0; r < p->rels.size(); ++r)
                if (!p->rels[r].s_set) SKF_FAIL(SKF_E_STATE, "backbone of relation %zu not set", r);
            if (!p->prepared) prepare_transform(p, st);
            for (int it = 0; it < n_iters; ++it) iterate_transform(p, st);
        } else {
            int it = 0;
            // The iteration is ~50-80 launches; for small graphs (launch-latency regime) the
            // remaining iterations replay ONE captured hipGraph.  Capture needs a real stream
            // (not the legacy default stream) and is skipped while profiling events are recorded.
            if (use_graph(p, st, n_iters)) {
                if (p->first_iter || !p->graph_exec || p->graph_stream != st) {
                    iterate_fit(p, st);               // eager: first-iteration work, attributes
                    ++it;
                    capture_iteration(p, st);
                }
                if (p->graph_exec)
                    for (; it < n_iters; ++it) SKF_HIP(hipGraphLaunch(p->graph_exec, st));
            }
            for (; it < n_iters; ++it) iterate_fit(p, st);
        }
    });
}

int skf_iterate_batch(skf_plan* const* plans, int32_t n_plans, int32_t n_iters, void* stream) {
    return guarded([&] {
        if (!plans || n_plans < 1 || n_plans > SKF_MAX_BATCH) SKF_FAIL(SKF_E_INVALID, "1 .. %d plans", SKF_MAX_BATCH);
        if (n_iters < 0) SKF_FAIL(SKF_E_INVALID, "n_iters < 0");
        skf_plan* p0 = plans[0];
        if (p0 && p0->variant == SKF_TRANSFORM) {
            // fold-ins of one graph into the models of several restarts (reference dfmf.py:191-199): same new relations, the
            // frozen factors / backbones of each restart in its own plan; one launch per iteration serves all of them
            for (int k = 0; k < n_plans; ++k) {
                skf_plan* p = plans[k];
                check_bound(p);
                if (!fold_fused(p) || p->dtype != p0->dtype || p->types.size() != p0->types.size() || p->rels.size() != p0->rels.size() ||
                    p->target != p0->target)
                    SKF_FAIL(SKF_E_STATE, "plan %d does not batch with plan 0 (fold-in without constraints on the target, same graph and engine required)", k);
                for (size_t i = 0; i < p->types.size(); ++i) {
                    if (p->types[i].n != p0->types[i].n || p->types[i].c != p0->types[i].c)
                        SKF_FAIL(SKF_E_STATE, "plan %d: object type %zu differs from plan 0", k, i);
                    if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "plan %d: factor of object type %zu not set", k, i);
                }
                for (size_t r = 0; r < p->rels.size(); ++r) {
                    if (p->rels[r].row != p0->rels[r].row || p->rels[r].col != p0->rels[r].col)
                        SKF_FAIL(SKF_E_STATE, "plan %d: relation %zu differs from plan 0", k, r);
                    if (!p->rels[r].s_set) SKF_FAIL(SKF_E_STATE, "plan %d: backbone of relation %zu not set", k, r);
                }
                for (int q = 0; q < k; ++q)
                    if (plans[q] == p) SKF_FAIL(SKF_E_INVALID, "plan %d listed twice", k);
            }
            hipStream_t st = as_stream(stream);
            for (int k = 0; k < n_plans; ++k)
                if (!plans[k]->prepared) prepare_transform(plans[k], st);
            fold_steps(plans, n_plans, n_iters, st);
            return;
        }
        for (int k = 0; k < n_plans; ++k) {
            skf_plan* p = plans[k];
            check_bound(p);
            for (size_t i = 0; i < p->types.size(); ++i)
                if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "plan %d: factor of object type %zu not set", k, i);
            // one launch serves every plan (plan 0's grid, LDS and job tables): the small-graph schedule, the same engine and
            // the same graph -- object counts, ranks, relation and constraint structure compared field by field
            bool same = p->small_fused && p->f64 == p0->f64 && p->variant == p0->variant && p->engine == p0->engine &&
                        p->types.size() == p0->types.size() && p->rels.size() == p0->rels.size() &&
                        p->thetas.size() == p0->thetas.size() && p->sm_j1.size() == p0->sm_j1.size() &&
                        p->sm_j3.size() == p0->sm_j3.size();
            for (size_t i = 0; same && i < p->types.size(); ++i)
                same = p->types[i].n == p0->types[i].n && p->types[i].c == p0->types[i].c;
            for (size_t r = 0; same && r < p->rels.size(); ++r)
                same = p->rels[r].row == p0->rels[r].row && p->rels[r].col == p0->rels[r].col;
            for (size_t t = 0; same && t < p->thetas.size(); ++t)
                same = p->thetas[t].type == p0->thetas[t].type && p->thetas[t].sparse == p0->thetas[t].sparse;
            auto same_job = [](const SmJob& a, const SmJob& b) {
                return a.kind == b.kind && a.idx == b.idx && a.r0 == b.r0 && a.nr == b.nr && a.part == b.part && a.k0 == b.k0 && a.nk == b.nk;
            };
            for (size_t j = 0; same && j < p->sm_j1.size(); ++j) same = same_job(p->sm_j1[j], p0->sm_j1[j]);
            for (size_t j = 0; same && j < p->sm_j3.size(); ++j) same = same_job(p->sm_j3[j], p0->sm_j3[j]);
            if (!same)
                SKF_FAIL(SKF_E_STATE, "plan %d does not batch with plan 0 (small-graph schedule, same graph and engine required)", k);
            for (int q = 0; q < k; ++q)
                if (plans[q] == p) SKF_FAIL(SKF_E_INVALID, "plan %d listed twice", k);
        }
        hipStream_t st = as_stream(stream);
        std::vector<const void*> tabs((size_t)n_plans);
        for (int k = 0; k < n_plans; ++k) tabs[(size_t)k] = plans[k]->sm_tables.ptr;
        if (tabs != p0->sm_batch_host) {                  // (the table of tables lives in plan 0's workspace; [0] stays plan 0's own)
            p0->sm_batch_host = tabs;
            SKF_HIP(hipMemcpyAsync(p0->sm_batch.ptr, p0->sm_batch_host.data(), tabs.size() * sizeof(void*), hipMemcpyHostToDevice, st));
        }
        for (int it = 0; it < n_iters; ++it) {
            if (p0->f64) iterate_small_fused_t<double>(p0, st, (unsigned)n_plans);
            else iterate_small_fused_t<float>(p0, st, (unsigned)n_plans);
        }
        for (int k = 0; k < n_plans; ++k) plans[k]->first_iter = false;
    });
}

int skf_plan_batchable(const skf_plan* p, int32_t* yes) {
    return guarded([&] {
        check_bound(p);
        if (!yes) SKF_FAIL(SKF_E_INVALID, "null pointer");
        *yes = (p->small_fused || fold_fused(p)) ? 1 : 0;
    });
}

int skf_small_graph_limits(int32_t* max_rank, int64_t* max_objects, int32_t* max_types, int32_t* max_relations,
                           int32_t* max_constraints, int32_t* constraint_nnz_divisor) {
    return guarded([&] {
        if (max_rank) *max_rank = SMALLC;
        if (max_objects) *max_objects = SM_MAX_OBJECTS;
        if (max_types) *max_types = SM_MAXT;
        if (max_relations) *max_relations = SM_MAXR;
        if (max_constraints) *max_constraints = SM_MAXTH;
        if (constraint_nnz_divisor) *constraint_nnz_divisor = SKF_THETA_SPARSE_DIV;
    });
}

int skf_plan_set_graph(skf_plan* p, int32_t enable) {
    return guarded([&] {
        if (!p) SKF_FAIL(SKF_E_INVALID, "null plan");
        p->graph_on = enable != 0;
    });
}

int skf_accumulate(skf_plan* p, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "skf_accumulate: SKF_DFMF / SKF_DFMC plans only");
        if (p->owned) SKF_FAIL(SKF_E_STATE, "a plan with owned rows (SKF_OPT_OWNED_ROWS) iterates through skf_iterate_dist");
        for (size_t i = 0; i < p->types.size(); ++i)
            if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "factor of object type %zu not set", i);
        accumulate_fit(p, as_stream(stream));
    });
}

int skf_apply_update(skf_plan* p, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "skf_apply_update: SKF_DFMF / SKF_DFMC plans only");
        apply_update(p, as_stream(stream));
    });
}

int skf_stage(skf_plan* p, int32_t stage, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "skf_stage: SKF_DFMF / SKF_DFMC plans only");
        if (p->owned) SKF_FAIL(SKF_E_STATE, "a plan with owned rows (SKF_OPT_OWNED_ROWS) iterates through skf_iterate_dist");
        for (size_t i = 0; i < p->types.size(); ++i)
            if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "factor of object type %zu not set", i);
        hipStream_t st = as_stream(stream);
        switch (stage) {
            case SKF_STAGE_CONTRACT: stage_contract(p, st); break;
            case SKF_STAGE_BACKBONE: stage_backbone(p, st); break;
            case SKF_STAGE_ACCUMULATE: stage_accumulate(p, st); break;
            case SKF_STAGE_UPDATE: apply_update(p, st); break;
            default: SKF_FAIL(SKF_E_INVALID, "unknown stage %d", stage);
        }
    });
}

int skf_exchange_range(const skf_plan* p, int32_t which, size_t* offset, size_t* bytes, int32_t* dtype) {
    return guarded([&] {
        if (!p || !offset || !bytes || !dtype) SKF_FAIL(SKF_E_INVALID, "null argument");
        *dtype = p->mt;
        switch (which) {
            case SKF_X_W: *offset = p->xw_off; *bytes = p->xw_bytes; *dtype = SKF_F64; break;
            case SKF_X_Q: *offset = p->xq_off; *bytes = p->xq_bytes; break;
            case SKF_X_QM: *offset = p->xqm_off; *bytes = p->xqm_bytes; break;
            case SKF_X_ED: *offset = p->acc_off; *bytes = p->acc_bytes; break;
            default: SKF_FAIL(SKF_E_INVALID, "unknown exchange range %d", which);
        }
    });
}

int skf_accumulator_range(const skf_plan* p, size_t* offset, size_t* bytes) {
    return guarded([&] {
        if (!p || !offset || !bytes) SKF_FAIL(SKF_E_INVALID, "null argument");
        *offset = p->acc_off;
        *bytes = p->acc_bytes;
    });
}

int skf_comm_unique_id(void* id128) {
    return guarded([&] {
        if (!id128) SKF_FAIL(SKF_E_INVALID, "null argument");
        NcclUniqueId id;
        const int rc = rccl().get_unique_id(&id);
        if (rc != 0) SKF_FAIL(SKF_E_HIP, "ncclGetUniqueId failed (%d)", rc);
        memcpy(id128, &id, sizeof id);
    });
}

int skf_comm_create(const void* id128, int32_t rank, int32_t world, skf_comm** out) {
    return guarded([&] {
        if (!out || world < 1 || rank < 0 || rank >= world) SKF_FAIL(SKF_E_INVALID, "bad rank / world / pointer");
        skf_comm* c = new skf_comm();
        c->rank = rank; c->world = world;
        if (id128) {
            NcclUniqueId id;
            memcpy(&id, id128, sizeof id);
            const int rc = rccl().comm_init_rank(&c->nccl, world, id, rank);
            if (rc != 0) {
                delete c;
                SKF_FAIL(SKF_E_HIP, "ncclCommInitRank failed: %s", g_rccl.error_string ? g_rccl.error_string(rc) : "?");
            }
        } else if (world != 1) {
            delete c;
            SKF_FAIL(SKF_E_INVALID, "a communicator of %d ranks needs the unique id of skf_comm_unique_id", world);
        }
        *out = c;
    });
}

int skf_comm_create_callback(int32_t rank, int32_t world, skf_collective_fn fn, void* user, skf_comm** out) {
    return guarded([&] {
        if (!out || !fn || world < 1 || rank < 0 || rank >= world) SKF_FAIL(SKF_E_INVALID, "bad rank / world / pointer");
        skf_comm* c = new skf_comm();
        c->rank = rank; c->world = world; c->fn = fn; c->user = user;
        *out = c;
    });
}

int skf_comm_create_null(int32_t rank, int32_t world, skf_comm** out) {
    return guarded([&] {
        if (!out || world < 1 || rank < 0 || rank >= world) SKF_FAIL(SKF_E_INVALID, "bad rank / world / pointer");
        skf_comm* c = new skf_comm();
        c->rank = rank; c->world = world; c->null_comm = true;
        *out = c;
    });
}

int skf_owned_rows(int32_t dtype, int64_t n_obj, int32_t part_index, int32_t part_count, int64_t* begin, int64_t* count,
                   int64_t* chunk) {
    return guarded([&] {
        if (n_obj <= 0 || part_count < 1 || part_index < 0 || part_index >= part_count || !begin || !count || !chunk)
            SKF_FAIL(SKF_E_INVALID, "bad argument");
        if (dtype != SKF_F64 && dtype != SKF_F32 && dtype != SKF_BF16) SKF_FAIL(SKF_E_INVALID, "unknown dtype %d", dtype);
        const int64_t ch = owned_chunk(dtype, n_obj, part_count);
        int64_t lo = ch * part_index, hi = lo + ch;
        if (lo > n_obj) lo = n_obj;
        if (hi > n_obj) hi = n_obj;
        *begin = lo;
        *count = hi - lo;
        *chunk = ch;
    });
}

int skf_abi_version(void) { return SKF_ABI_VERSION; }

int skf_comm_info(const skf_comm* c, int32_t* rank, int32_t* world, int32_t* transport, int32_t* transport_ranks) {
    return guarded([&] {
        if (!c) SKF_FAIL(SKF_E_INVALID, "null communicator");
        if (rank) *rank = c->rank;
        if (world) *world = c->world;
        const int kind = c->nccl ? SKF_COMM_RCCL : c->fn ? SKF_COMM_CALLBACK : c->null_comm ? SKF_COMM_NULL : SKF_COMM_SINGLE;
        if (transport) *transport = kind;
        if (transport_ranks) {
            *transport_ranks = kind == SKF_COMM_SINGLE ? 1 : kind == SKF_COMM_NULL ? 0 : c->world;
            if (c->nccl) {                     // what RCCL itself says about the communicator it built
                int n = -1, r = -1;
                if (g_rccl.comm_count && g_rccl.comm_count(c->nccl, &n) == 0) *transport_ranks = n;
                else *transport_ranks = -1;
                if (g_rccl.comm_user_rank && g_rccl.comm_user_rank(c->nccl, &r) == 0 && r != c->rank)
                    SKF_FAIL(SKF_E_STATE, "RCCL numbers this rank %d, the communicator was created as rank %d", r, c->rank);
            }
        }
    });
}

int skf_launch_count(int64_t* launches) {
    return guarded([&] {
        if (!launches) SKF_FAIL(SKF_E_INVALID, "null pointer");
        *launches = g_launches;
    });
}

int skf_comm_destroy(skf_comm* c) {
    return guarded([&] {
        if (!c) return;
        if (c->nccl && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->nccl);
        delete c;
    });
}

int skf_plan_set_comm(skf_plan* p, skf_comm* comm) {
    return guarded([&] {
        if (!p) SKF_FAIL(SKF_E_INVALID, "null plan");
        if (p->variant == SKF_TRANSFORM && comm) SKF_FAIL(SKF_E_INVALID, "skf_plan_set_comm: SKF_DFMF / SKF_DFMC plans only");
        p->comm = comm;
    });
}

int skf_iterate_dist(skf_plan* p, int32_t n_iters, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (p->variant == SKF_TRANSFORM) SKF_FAIL(SKF_E_INVALID, "skf_iterate_dist: SKF_DFMF / SKF_DFMC plans only");
        if (!p->comm) SKF_FAIL(SKF_E_STATE, "no communicator attached to the plan (skf_plan_set_comm)");
        if (n_iters < 0) SKF_FAIL(SKF_E_INVALID, "n_iters < 0");
        for (size_t i = 0; i < p->types.size(); ++i)
            if (!p->types[i].set) SKF_FAIL(SKF_E_STATE, "factor of object type %zu not set", i);
        if (p->owned) {
            if (p->comm->world != p->part_count || p->comm->rank != p->part_index)
                SKF_FAIL(SKF_E_STATE, "the communicator is rank %d of %d, the plan part %d of %d", p->comm->rank, p->comm->world,
                         p->part_index, p->part_count);
            for (int it = 0; it < n_iters; ++it) iterate_owned(p, as_stream(stream));
            finalize_owned(p, as_stream(stream));
            return;
        }
        for (int it = 0; it < n_iters; ++it) iterate_dist(p, as_stream(stream));
    });
}

int skf_exchange_bytes(const skf_plan* p, int32_t world, size_t* bytes) {
    return guarded([&] {
        if (!p || !bytes || world < 1) SKF_FAIL(SKF_E_INVALID, "bad argument");
        *bytes = exchange_bytes(p, world);
    });
}

int skf_relation_sqerr(skf_plan* p, int32_t rel, double* out, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (rel < 0 || rel >= (int)p->rels.size() || !out) SKF_FAIL(SKF_E_INVALID, "bad relation index / pointer");
        hipStream_t st = as_stream(stream);
        RelState& r = p->rels[rel];
        TypeState& ti = p->types[r.row];
        TypeState& tj = p->types[r.col];
        const int ni = (int)r.nr, nj = (int)tj.n, ci = ti.c, cj = tj.c;        // the local rows
        if (r.absent) {
            SKF_HIP(hipMemsetAsync(out, 0, sizeof(double), st));
            return;
        }
        if (r.kn) {
            // The completed relation of _dfmc.py:385-386 is X_o + E (X_o = G_i,prev S_prev G_j,prev^T, E the stored residuals);
            // with X_n = G_i S G_j^T of the current factors
            //     |R_c - X_n|^2 = |X_o - X_n|^2 + sum over the known entries of (r - x_n)^2 - (x_o - x_n)^2 ,  x_o = r - e
            // the first term from c x c Gram / cross-Gram products, the second from one pass over the column lists.
            double* acc = (double*)p->sqpart.ptr;
            auto trace_term = [&](const void* Ai, const void* S1, const void* Bj, const void* S2, double scale, bool first) {
                GemmArgs h = gemm_args(Ai, ci, 1, S1, cj, 1, r.U.ptr, cj, ci, cj, ci, EPI_STORE, 0);          // U = A_i S1
                small_gemm(p, h, st);
                h = gemm_args(r.U.ptr, cj, 1, Bj, cj, 1, r.T1.ptr, cj, ci, cj, cj, EPI_STORE, 0);             // T1 = U B_j
                small_gemm(p, h, st);
                hipLaunchKernelGGL(dot_small_kernel, dim3(1), dim3(256), 0, st, (const double*)S2, (const double*)r.T1.ptr,
                                   (int64_t)ci * cj, scale, acc, first ? 0 : 1);
                check_launch("dot_small");
            };
            auto gram_of = [&](const void* A, const void* B, void* C, int c, int64_t n) {                      // C = A^T B (c x c)
                GemmArgs h = gemm_args(A, 1, c, B, c, 1, C, c, c, c, (int)n, EPI_STORE, 0);
                wide_gemm(p, h, st);
            };
            // (the partial slots [1, waves] belong to the pass below; slot 0 collects the trace terms)
            const void* Gi_b = rows_of(p, ti.G, ti, r.r0);                   // the rows of the block (row ownership: the local ones)
            const void* Gp_b = ti.Gp.ptr ? rows_of(p, ti.Gp, ti, r.r0) : nullptr;
            gram_of(Gi_b, Gi_b, r.Xi.ptr, ci, r.nr);
            gram_of(tj.G.ptr, tj.G.ptr, r.Xj.ptr, cj, tj.n);
            trace_term(r.Xi.ptr, r.S.ptr, r.Xj.ptr, r.S.ptr, 1.0, true);
            if (!p->kn_first) {
                gram_of(Gp_b, Gi_b, r.Xi.ptr, ci, r.nr);                     // G_i,prev^T G_i
                gram_of(tj.G.ptr, tj.Gp.ptr, r.Xj.ptr, cj, tj.n);            // G_j^T G_j,prev
                trace_term(r.Xi.ptr, r.S.ptr, r.Xj.ptr, r.Sp.ptr, -2.0, false);
                gram_of(Gp_b, Gp_b, r.Xi.ptr, ci, r.nr);
                gram_of(tj.Gp.ptr, tj.Gp.ptr, r.Xj.ptr, cj, tj.n);
                trace_term(r.Xi.ptr, r.Sp.ptr, r.Xj.ptr, r.Sp.ptr, 1.0, false);
            }
            GemmArgs g2 = gemm_args(tj.G.ptr, cj, 1, r.S.ptr, 1, cj, r.Tm.ptr, ci, nj, ci, cj, EPI_STORE, 0);  // T = G_j S^T
            mixed_gemm(p, g2, st);
            if (p->bf16) launch_to_bf16<float>((uint16_t*)r.FiB.ptr, r.kn_ldf, (const float*)r.Tm.ptr, (int64_t)ci, tj.n, ci, false, st);
            const int waves = known_pass(p, r, true, SRP_ERR, st, 1);       // the pass writes its partials behind slot 0
                                                                            // (capacity checked before it launches)
            hipLaunchKernelGGL((sum_partials_kernel<double>), dim3(1), dim3(256), 0, st, (const double*)p->sqpart.ptr, waves + 1, out);
            check_launch("sum_partials");
            return;
        }
        GemmArgs g = gemm_args(rows_of(p, ti.G, ti, r.r0), ci, 1, r.S.ptr, cj, 1, r.H.ptr, cj, ni, cj, ci, EPI_STORE, 0);
        mixed_gemm(p, g, st);
        if (p->bf16) {
            // one pass over the stored bf16 relation: bf16 H and G_j on the matrix cores, f32 residual
            launch_tile_epilogue(p, r, MODE_SQERR, st);
            hipLaunchKernelGGL((sum_partials_kernel<double>), dim3(1), dim3(256), 0, st, (const double*)p->sqpart.ptr,
                               cdiv(ni, 256) * cdiv(nj, 256), out);
            check_launch("sum_partials");
            return;
        }
        g = gemm_args(r.H.ptr, cj, 1, tj.G.ptr, 1, cj, (void*)r.R, r.ldr, ni, nj, cj, EPI_SQDIFF, 0);
        g.C2 = p->sqpart.ptr;
        // one partial per workgroup of the tile run_gemm picks for THIS product (an all-f64 product of a small
        // relation with 64 <= c_j <= 1024 runs on the deep 32 x 32 tile)
        const TileCfg t = gemm_tile(GemmTypes{p->mt, p->mt, p->mt}, p->engine, g, p->f64, false);
        const int blocks = cdiv(ni, t.bm) * cdiv(nj, t.bn);
        if ((size_t)blocks > p->sq_elems) SKF_FAIL(SKF_E_STATE, "residual partials: %d tiles > %zu slots", blocks, p->sq_elems);
        plan_gemm(p, g, st);
        if (p->f64)
            hipLaunchKernelGGL((sum_partials_kernel<double>), dim3(1), dim3(256), 0, st, (const double*)p->sqpart.ptr,
                               blocks, out);
        else
            hipLaunchKernelGGL((sum_partials_kernel<float>), dim3(1), dim3(256), 0, st, (const float*)p->sqpart.ptr,
                               blocks, out);
        check_launch("sum_partials");
    });
}

int skf_get_contraction(const skf_plan* p, int32_t rel, int32_t which, void* dst, int64_t ld, void* stream) {
    return guarded([&] {
        check_bound(p);
        if (rel < 0 || rel >= (int)p->rels.size() || !dst || which < 0 || which > 2)
            SKF_FAIL(SKF_E_INVALID, "bad relation index / selector / pointer");
        const RelState& r = p->rels[rel];
        const TypeState& ti = p->types[r.row];
        const TypeState& tj = p->types[r.col];
        if (which == 2 && !r.kn) SKF_FAIL(SKF_E_STATE, "relation %d forms P, not P S^T (which = 2 is for known-entries relations)", rel);
        const Slot& src = which == 0 ? r.P : which == 1 ? r.Q : r.A;
        const int64_t rows = which == 1 ? tj.n : r.nr, cols = which == 0 ? tj.c : ti.c;
        if (p->small_fused && which == 1 && r.Q.ptr) {       // the fused small-graph schedule keeps Q as shares: sum them first
            const int64_t total = rows * cols;
            if (p->f64)
                hipLaunchKernelGGL((sum_parts_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, as_stream(stream), (double*)r.Q.ptr,
                                   (const double*)r.SmQ.ptr, total, r.sm_qparts, total);
            else
                hipLaunchKernelGGL((sum_parts_kernel<float>), dim3(elem_grid(total)), dim3(256), 0, as_stream(stream), (float*)r.Q.ptr,
                                   (const float*)r.SmQ.ptr, total, r.sm_qparts, total);
            check_launch("sum_parts");
        }
        if (!src.ptr || rows <= 0) SKF_FAIL(SKF_E_STATE, "relation %d keeps no %s here", rel, which == 0 ? "P" : "Q");
        if (ld < cols) SKF_FAIL(SKF_E_INVALID, "ld too small");
        copy2d(dst, ld, src.ptr, cols, rows, cols, p->esz, as_stream(stream));
    });
}

int skf_plan_set_profiling(skf_plan* p, int32_t enable) {
    return guarded([&] {
        if (!p) SKF_FAIL(SKF_E_INVALID, "null plan");
        p->profiling = enable != 0;
        p->ev_used = 0;
        p->prof_flops = p->prof_bytes = 0.0;
        p->prof_launches = 0;
    });
}

int skf_plan_get_profile(skf_plan* p, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    return guarded([&] {
        if (!p || !total_ms || !launches || !flops || !bytes) SKF_FAIL(SKF_E_INVALID, "null argument");
        double ms = 0.0;
        for (size_t k = 0; k + 1 < p->ev_used; k += 2) {
            SKF_HIP(hipEventSynchronize(p->ev_pool[k + 1]));
            float t = 0.f;
            SKF_HIP(hipEventElapsedTime(&t, p->ev_pool[k], p->ev_pool[k + 1]));
            ms += t;
        }
        *total_ms = ms;
        *launches = p->prof_launches;
        *flops = p->prof_flops;
        *bytes = p->prof_bytes;
        p->ev_used = 0;
        p->prof_flops = p->prof_bytes = 0.0;
        p->prof_launches = 0;
    });
}

int skf_gemm(int32_t dtype, int32_t engine, const skf_gemm_desc* d, void* workspace, size_t workspace_bytes,
             void* stream) {
    return guarded([&] {
        if (!d || !d->A || !d->B || !d->C) SKF_FAIL(SKF_E_INVALID, "null argument");
        if (dtype != SKF_F64 && dtype != SKF_F32) SKF_FAIL(SKF_E_INVALID, "skf_gemm: dtype must be SKF_F64 / SKF_F32");
        if (d->epi < EPI_STORE || d->epi > EPI_MASKED_STORE) SKF_FAIL(SKF_E_INVALID, "bad epilogue");
        if ((d->epi == EPI_SPLIT_STORE || d->epi == EPI_SPLIT_ACC) && !d->C2) SKF_FAIL(SKF_E_INVALID, "C2 missing");
        if (d->epi == EPI_MASKED_STORE && !d->mask) SKF_FAIL(SKF_E_INVALID, "mask missing");
        if (d->M < 0 || d->N < 0 || d->K < 0) SKF_FAIL(SKF_E_INVALID, "negative dimension");
        GemmArgs g = gemm_args(d->A, d->sa_m, d->sa_k, d->B, d->sb_k, d->sb_n, d->C, d->ldc, d->M, d->N, d->K,
                               d->epi, d->nan_to_num);
        g.C2 = d->C2; g.ldc2 = d->ldc2 ? d->ldc2 : d->ldc;
        g.mask = d->mask; g.ldmask = d->ldmask;
        g.aop = d->aop;
        GemmTypes ty{dtype, d->a_dtype < 0 ? dtype : d->a_dtype, d->b_dtype < 0 ? dtype : d->b_dtype};
        // X^T X (one operand read both ways, plain store): a symmetric product -- the split-K form computes the tiles on / below
        // the diagonal only (GemmArgs::sym; bit for bit the full product)
        static const int sym_on = env_int("SKF_GRAM_SYM", 1) != 0;
        if (sym_on && d->A == d->B && d->M == d->N && d->sa_m == d->sb_n && d->sa_k == d->sb_k && d->epi == EPI_STORE &&
            d->aop == AOP_NONE && ty.a == ty.b)
            g.sym = 1;
        run_gemm(ty, engine, g, d->splits, workspace, workspace_bytes, as_stream(stream));
    });
}

int skf_gemm_bf16(const void* A, int64_t lda, const void* Bt, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                  int32_t Kp, int32_t splits, void* workspace, size_t workspace_bytes, void* stream) {
    return guarded([&] {
        if (!A || !Bt || !C || M < 0 || N < 0 || Kp < 0 || ldc < N) SKF_FAIL(SKF_E_INVALID, "bad argument");
        run_gemm_bf16((const uint16_t*)A, lda, (const uint16_t*)Bt, ldb, C, ldc, M, N, Kp, splits, workspace,
                      workspace_bytes, false, as_stream(stream));
    });
}

int skf_gemm_bf16_tn(const void* A, int64_t lda, const void* Bt, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                     int32_t Kp, int32_t splits, void* workspace, size_t workspace_bytes, void* stream) {
    return guarded([&] {
        if (!A || !Bt || !C || M < 0 || N < 0 || Kp < 0 || ldc < N) SKF_FAIL(SKF_E_INVALID, "bad argument");
        run_gemm_bf16((const uint16_t*)A, lda, (const uint16_t*)Bt, ldb, C, ldc, M, N, Kp, splits, workspace,
                      workspace_bytes, false, as_stream(stream), true);
    });
}

int skf_gemm_bits(const void* A, int64_t lda_bytes, const void* Bt, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                  int32_t Kp, int32_t transposed, int32_t splits, void* workspace, size_t workspace_bytes, void* stream) {
    return guarded([&] {
        if (!A || !Bt || !C || M < 0 || N < 0 || Kp < 0 || ldc < N) SKF_FAIL(SKF_E_INVALID, "bad argument");
        run_gemm_bf16((const uint16_t*)A, lda_bytes, (const uint16_t*)Bt, ldb, C, ldc, M, N, Kp, splits, workspace,
                      workspace_bytes, false, as_stream(stream), transposed != 0, true);
    });
}

int skf_to_bf16(void* dst, int64_t ldd, int32_t src_dtype, const void* src, int64_t lds, int64_t rows, int64_t cols,
                int32_t transpose, void* stream) {
    return guarded([&] {
        if (!dst || !src || rows < 0 || cols < 0) SKF_FAIL(SKF_E_INVALID, "bad argument");
        hipStream_t st = as_stream(stream);
        if (src_dtype == SKF_F64) launch_to_bf16<double>((uint16_t*)dst, ldd, (const double*)src, lds, rows, cols, transpose != 0, st);
        else if (src_dtype == SKF_F32) launch_to_bf16<float>((uint16_t*)dst, ldd, (const float*)src, lds, rows, cols, transpose != 0, st);
        else if (src_dtype == SKF_BF16) launch_to_bf16<uint16_t>((uint16_t*)dst, ldd, (const uint16_t*)src, lds, rows, cols, transpose != 0, st);
        else SKF_FAIL(SKF_E_INVALID, "bad source dtype");
    });
}

int skf_pinv_sym_workspace_bytes(int32_t n, size_t* bytes) {
    return guarded([&] {
        if (n <= 0 || n > EIGH_MAXN - 1 || !bytes) SKF_FAIL(SKF_E_INVALID, "bad order");
        const size_t np = (size_t)(n + 1) / 2 * 2;
        *bytes = align_up(np * np * 8, 256) * 3 + align_up(np * 8, 256) + 512;
        if (n > SWEEP_MAXN) *bytes += defl_scratch_bytes(1, (int64_t)np * np) + 256;
    });
}

int skf_pinv_sym(int32_t dtype, const void* A, int64_t lda, void* K, int64_t ldk, int32_t n, void* ws,
                 size_t ws_bytes, void* stream) {
    return guarded([&] {
        size_t need = 0;
        if (skf_pinv_sym_workspace_bytes(n, &need) != SKF_OK) SKF_FAIL(SKF_E_INVALID, "bad order %d", n);
        if (!A || !K || !ws || ws_bytes < need) SKF_FAIL(SKF_E_WORKSPACE, "pinv workspace too small / null pointer");
        if (dtype != SKF_F64 && dtype != SKF_F32) SKF_FAIL(SKF_E_INVALID, "bad dtype");
        hipStream_t st = as_stream(stream);
        const int np = (n + 1) / 2 * 2;
        const size_t mat = align_up((size_t)np * np * 8, 256);
        char* base = (char*)ws;
        double* eA = (double*)base;
        double* eV = (double*)(base + mat);
        double* eVs = (double*)(base + 2 * mat);
        double* eW = (double*)(base + 3 * mat);
        int* eN = (int*)(base + 3 * mat + align_up((size_t)np * 8, 256));
        int* eNo = eN + 16;
        int* eOk = eN + 32;
        const int total = np * np;
        if (dtype == SKF_F64)
            hipLaunchKernelGGL((eigh_pack_kernel<double>), dim3(elem_grid(total)), dim3(256), 0, st, eA, np,
                               (const double*)A, lda, n, eN, eNo);
        else
            hipLaunchKernelGGL((eigh_pack_kernel<float>), dim3(elem_grid(total)), dim3(256), 0, st, eA, np,
                               (const float*)A, lda, n, eN, eNo);
        check_launch("eigh_pack");
        EighArgs e;
        e.A = eA; e.V = eV; e.Vs = eVs; e.w = eW; e.stride = (int64_t)np * np; e.wstride = np;
        e.n = eN; e.n_orig = eNo; e.chol_ok = eOk; e.max_sweeps = 30;
        const int tot2 = n * n;
        const Switches sw = Switches::read();          // stand-alone operator: no plan to hold them
        // fast path: the blocked sweep for orders 65 .. 256 (writes a contiguous f64 K itself), else Cholesky inverse + unpack
        const bool sweep = dtype == SKF_F64 && ldk == n && sweep_takes(sw, n);
        if (sweep) {
            PinvBatch pb;
            memset(&pb, 0, sizeof pb);
            pb.K[0] = (double*)K; pb.c[0] = n; pb.n_pad[0] = np;
            launch_sweep(sw, e, pb, 1, n, st);
        } else {
            launch_chol(sw, e, 1, np, st);
            if (dtype == SKF_F64)
                hipLaunchKernelGGL((chol_unpack_kernel<double>), dim3(elem_grid(tot2)), dim3(256), 0, st, (double*)K, ldk,
                                   eV, np, n, eOk);
            else
                hipLaunchKernelGGL((chol_unpack_kernel<float>), dim3(elem_grid(tot2)), dim3(256), 0, st, (float*)K, ldk,
                                   eV, np, n, eOk);
            check_launch("chol_unpack");
        }
        if (sweep && defl_multi_takes(sw, n)) {      // orders above 256: the deflation over several workgroups first
            PinvBatch pb;
            memset(&pb, 0, sizeof pb);
            pb.K[0] = (double*)K; pb.c[0] = n; pb.n_pad[0] = np;
            char* scratch = base + align_up(mat * 3 + align_up((size_t)np * 8, 256) + 512, 256);
            launch_deflation_multi(sw, SKF_ENGINE_MFMA, e, pb, 1, n, scratch, st);
        }
        {
            static DeviceOnce once;
            allow_dynamic_lds(once, pchol_pinv_kernel, PCHOL_LDS_BYTES);
        }
        const int lr = np < PCHOL_LDS_R ? np : PCHOL_LDS_R;
        hipLaunchKernelGGL(pchol_pinv_kernel, dim3(1), dim3(EIGH_THREADS), (size_t)lr * (lr + 1) / 2 * 8, st, e, deflation_lo(sw), 1e-7, lr);
        check_launch("pchol_pinv");
        hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(EIGH_THREADS), 0, st, e);
        check_launch("jacobi_eigh");
        if (dtype == SKF_F64)
            hipLaunchKernelGGL((eigh_unpack_pinv_kernel<double>), dim3(elem_grid(tot2)), dim3(256), 0, st, (double*)K,
                               ldk, eVs, eV, np, n, eOk);
        else
            hipLaunchKernelGGL((eigh_unpack_pinv_kernel<float>), dim3(elem_grid(tot2)), dim3(256), 0, st, (float*)K,
                               ldk, eVs, eV, np, n, eOk);
        check_launch("eigh_unpack");
    });
}

int skf_fill_uniform(int32_t dtype, void* dst, int64_t rows, int64_t cols, int64_t ld, uint64_t seed, double scale,
                     double shift, void* stream) {
    return guarded([&] {
        if (!dst || rows < 0 || cols < 0 || ld < cols) SKF_FAIL(SKF_E_INVALID, "bad argument");
        hipStream_t st = as_stream(stream);
        const int grid = elem_grid(rows * cols);
        if (dtype == SKF_F64)
            hipLaunchKernelGGL((fill_uniform_kernel<double>), dim3(grid), dim3(256), 0, st, (double*)dst, rows, cols, ld,
                               seed, scale, shift);
        else if (dtype == SKF_F32)
            hipLaunchKernelGGL((fill_uniform_kernel<float>), dim3(grid), dim3(256), 0, st, (float*)dst, rows, cols, ld,
                               seed, scale, shift);
        else if (dtype == SKF_BF16)
            hipLaunchKernelGGL((fill_uniform_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, (uint16_t*)dst, rows, cols,
                               ld, seed, scale, shift);
        else
            SKF_FAIL(SKF_E_INVALID, "bad dtype");
        check_launch("fill_uniform");
    });
}

int skf_fill_unknown_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes) {
    return guarded([&] {
        if (rows < 0 || cols < 0 || !bytes) SKF_FAIL(SKF_E_INVALID, "bad argument");
        *bytes = (size_t)(2 * rows + 2 * cols + 2) * sizeof(double);
    });
}

int skf_fill_unknown(int32_t dtype, void* data, int64_t ld, int64_t rows, int64_t cols, const uint8_t* mask, int64_t mask_ld,
                     int32_t strategy, double value, void* workspace, size_t workspace_bytes, void* stream) {
    return guarded([&] {
        if (!data || rows < 0 || cols < 0 || ld < cols || (mask && mask_ld < cols)) SKF_FAIL(SKF_E_INVALID, "bad argument");
        if (dtype != SKF_F64 && dtype != SKF_F32) SKF_FAIL(SKF_E_INVALID, "skf_fill_unknown: dtype must be SKF_F64 / SKF_F32");
        if (strategy < FILL_MEAN || strategy > FILL_CONST) SKF_FAIL(SKF_E_INVALID, "unknown fill strategy %d", strategy);
        size_t need = 0;
        skf_fill_unknown_workspace_bytes(rows, cols, &need);
        if (!workspace || workspace_bytes < need) SKF_FAIL(SKF_E_WORKSPACE, "fill workspace too small / null");
        if (rows == 0 || cols == 0) return;
        hipStream_t st = as_stream(stream);
        double* stats = (double*)workspace;
        const int rgrid = (int)((rows + 3) / 4 < 2048 ? ((rows + 3) / 4 > 0 ? (rows + 3) / 4 : 1) : 2048);     // 4 waves per workgroup
        if (dtype == SKF_F64) {
            double* X = (double*)data;
            if (strategy != FILL_CONST) {
                hipLaunchKernelGGL((fill_row_stats_kernel<double>), dim3(rgrid), dim3(256), 0, st, X, ld, rows, cols, mask, mask_ld, stats);
                if (strategy == FILL_COL_MEAN)
                    hipLaunchKernelGGL((fill_col_stats_kernel<double>), dim3(elem_grid(cols)), dim3(256), 0, st, X, ld, rows, cols, mask, mask_ld, stats);
                hipLaunchKernelGGL(fill_total_kernel, dim3(1), dim3(256), 0, st, rows, cols, stats);
            }
            hipLaunchKernelGGL((fill_apply_kernel<double>), dim3(elem_grid(rows * cols)), dim3(256), 0, st, X, ld, rows, cols, mask,
                               mask_ld, stats, strategy, value);
        } else {
            float* X = (float*)data;
            if (strategy != FILL_CONST) {
                hipLaunchKernelGGL((fill_row_stats_kernel<float>), dim3(rgrid), dim3(256), 0, st, X, ld, rows, cols, mask, mask_ld, stats);
                if (strategy == FILL_COL_MEAN)
                    hipLaunchKernelGGL((fill_col_stats_kernel<float>), dim3(elem_grid(cols)), dim3(256), 0, st, X, ld, rows, cols, mask, mask_ld, stats);
                hipLaunchKernelGGL(fill_total_kernel, dim3(1), dim3(256), 0, st, rows, cols, stats);
            }
            hipLaunchKernelGGL((fill_apply_kernel<float>), dim3(elem_grid(rows * cols)), dim3(256), 0, st, X, ld, rows, cols, mask,
                               mask_ld, stats, strategy, value);
        }
        check_launch("fill_unknown");
    });
}

int skf_cast(int32_t dst_dtype, void* dst, int64_t ldd, int32_t src_dtype, const void* src, int64_t lds, int64_t rows,
             int64_t cols, void* stream) {
    return guarded([&] {
        if (!dst || !src || rows < 0 || cols < 0) SKF_FAIL(SKF_E_INVALID, "bad argument");
        hipStream_t st = as_stream(stream);
        const int grid = elem_grid(rows * cols);
        if (dst_dtype == SKF_F32 && src_dtype == SKF_F64)
            hipLaunchKernelGGL((cast_kernel<float, double>), dim3(grid), dim3(256), 0, st, (float*)dst, ldd,
                               (const double*)src, lds, rows, cols);
        else if (dst_dtype == SKF_F64 && src_dtype == SKF_F32)
            hipLaunchKernelGGL((cast_kernel<double, float>), dim3(grid), dim3(256), 0, st, (double*)dst, ldd,
                               (const float*)src, lds, rows, cols);
        else if (dst_dtype == SKF_F32 && src_dtype == SKF_F32)
            hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (float*)dst, ldd,
                               (const float*)src, lds, rows, cols);
        else if (dst_dtype == SKF_F64 && src_dtype == SKF_F64)
            hipLaunchKernelGGL((cast_kernel<double, double>), dim3(grid), dim3(256), 0, st, (double*)dst, ldd,
                               (const double*)src, lds, rows, cols);
        else
            SKF_FAIL(SKF_E_INVALID, "unsupported cast %d -> %d", src_dtype, dst_dtype);
        check_launch("cast");
    });
}

}  // extern "C"
