// skf_known.h -- DFMC on the KNOWN entries only (sparse-residual form of reference _dfmc.py:287-292, 311-325, 341-352).
//
// The reference keeps a completed copy R_c of every masked relation: known entries = R, unknown entries = the
// reconstruction X = G_i S G_j^T of the current iteration, and contracts that dense matrix three times per iteration
// (W = G_i^T R_c G_j for the backbone, P = R_c G_j and Q = R_c^T G_i for the factor update).  With K = the known
// pattern and E = K o (R - X) (non-zero on the known entries only)
//     R_c = X + E = (G_i S) G_j^T + E
// so every product of R_c splits into c x c algebra on the factors plus a product of the SPARSE residual E:
//     P S^T = G_i (S Gram_j S^T) + E T ,  T = G_j S^T              row side of the factor update   (_dfmc.py:127-178)
//     Q     = G_j (S^T Gram_i)   + E^T G_i                         column side
//     W'    = (G_i'^T G_i) S (G_j^T G_j') + (E^T G_i')^T G_j'      backbone of the NEXT iteration (new factors G')
// and R_c is never written or read: work and traffic ~ nnz_known * c + n * c^2 instead of n_i * n_j * c.  The same
// associativity freedom as the 2-GEMM form of the dense path, and more accurate than rounding completed entries to the
// relation's storage type.
//
// Data: the known entries as CSR (rows -> ascending columns) and CSC (columns -> ascending rows), both with the R
// values, the CSC also with the residuals E of the last completed iteration.  Each list is cut into PARTS of the inner
// index (column parts of the row lists, row parts of the column lists) so that the slice of the gathered factor a
// part touches fits the 4 MiB L2 of ONE XCD; workgroup b handles part (b % 8) % parts, i.e. the parts are pinned to
// XCDs through the observed round-robin placement (speed only -- nothing depends on it).  The partial outputs of the
// parts are summed in a fixed order: results are run-to-run deterministic, no float atomics.
#pragma once
#include "skf_kernels.h"
#include <type_traits>

namespace skf {

enum { SRP_RESIDUAL = 0, SRP_APPLY = 1, SRP_ERR = 2 };

// One pass over the known entries, outer objects o (rows of the CSR or columns of the CSC), inner objects i:
//   SRP_RESIDUAL  x = <Fo[o], Fi[i]>, e = r - x, out[part][o] += e * Fi[i]   (e stored to evals when given)
//   SRP_APPLY     e = evals[q],                out[part][o] += e * Fi[i]
//   SRP_ERR       x' = <Fo[o], Fi[i]>, sq += (r - x')^2 - (r - e - x')^2      (e = evals[q]; one partial per wave)
template <typename TG, typename TM>
struct SrpArgs {
    const int64_t* ptr;         // [n_out * parts + 1]: segment (o, part) = [ptr[o * parts + part], ptr[o * parts + part + 1])
    const int* idx;             // inner index of every entry
    const TM* rvals;            // R at the entry
    TM* evals;                  // residuals (see above)
    const TG* Fo;               // [n_out][ldo]  vectors of the outer objects (width w)
    const TG* Fi;               // [n_in][ldi]   gathered vectors of the inner objects
    TM* out;                    // [parts][n_out][ld_out]
    double* sq;                 // SRP_ERR: [gridDim.x * 4] partials
    int64_t ldo, ldi, ld_out, part_stride, n_out;
    int w, parts, mode;
    uint32_t zero_off;          // srp_bf16_v6_kernel: byte offset of an all-zero row behind Fi (0 = none: srp_bf16_kernel runs)
};

template <typename TG> struct GatherT;
template <> struct GatherT<uint16_t> {          // bf16 bit patterns
    typedef float acc;
    static __device__ __forceinline__ float get(uint16_t v) { return bf16_to_f32(v); }
};
template <> struct GatherT<float> {
    typedef float acc;
    static __device__ __forceinline__ float get(float v) { return v; }
};
template <> struct GatherT<double> {
    typedef double acc;
    static __device__ __forceinline__ double get(double v) { return v; }
};

// workgroup -> (part, outer object of wave 0): blocks b, b + 8, ... share an XCD; part = (b % 8) % parts
__device__ __forceinline__ void srp_segment(int parts, int& part, int64_t& first) {
    const int xcd = blockIdx.x & 7;
    const int lg = parts >= 8 ? 3 : parts >= 4 ? 2 : parts >= 2 ? 1 : 0;      // parts is 1, 2, 4 or 8
    part = xcd & (parts - 1);
    first = ((int64_t)(blockIdx.x >> 3) * (8 >> lg) + (xcd >> lg)) * 4;       // 4 waves = 4 outer objects per workgroup
}

// 16-byte row chunks: GL lanes cover one gathered vector (w = GL * 16 / sizeof(TG)), 64 / GL entries per wave step.
// The entries of a segment are taken 64 at a time: one coalesced load of their indices / values (lane l holds entry
// l of the batch), then SRP_U wave steps' gathers are issued back to back before the first is consumed -- the pass is
// bound by the latency of the gathers (L2 / Infinity Cache hits), so what counts is the number of rows in flight.
constexpr int SRP_U = 4;
template <typename TG, typename TM, int GL>
__global__ __launch_bounds__(256) void srp_vec_kernel(SrpArgs<TG, TM> a) {
    constexpr int VE = 16 / (int)sizeof(TG);
    constexpr int EPW = 64 / GL;
    typedef TG vec_t __attribute__((ext_vector_type(VE)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane / GL, sub = lane % GL;
    int part;
    int64_t first;
    srp_segment(a.parts, part, first);
    const int64_t o = first + wv;
    double sq = 0.0;
    if (o < a.n_out) {                                  // (wave-uniform)
        const int64_t qa = a.ptr[o * a.parts + part], qb = a.ptr[o * a.parts + part + 1];
        TM fo[VE], acc[VE];
#pragma unroll
        for (int t = 0; t < VE; ++t) { fo[t] = (TM)0; acc[t] = (TM)0; }
        if (a.mode != SRP_APPLY) {
            const vec_t v = *(const vec_t*)(a.Fo + o * a.ldo + sub * VE);
#pragma unroll
            for (int t = 0; t < VE; ++t) fo[t] = GatherT<TG>::get(v[t]);
        }
        for (int64_t q0 = qa; q0 < qb; q0 += 64) {
            const int nb = (int)(qb - q0 < 64 ? qb - q0 : 64);           // entries of this batch
            const bool mine = lane < nb;
            const int my_i = mine ? a.idx[q0 + lane] : 0;
            const TM my_r = (mine && a.mode != SRP_APPLY) ? a.rvals[q0 + lane] : (TM)0;
            TM my_e = (mine && a.mode != SRP_RESIDUAL) ? a.evals[q0 + lane] : (TM)0;
            for (int s0 = 0; s0 < nb; s0 += EPW * SRP_U) {
                vec_t v[SRP_U];
                bool ok[SRP_U];
#pragma unroll
                for (int u = 0; u < SRP_U; ++u) {
                    const int ent = s0 + u * EPW + grp;
                    const int i = __shfl(my_i, ent & 63, 64);
                    ok[u] = ent < nb;
#pragma unroll
                    for (int t = 0; t < VE; ++t) v[u][t] = (TG)0;
                    if (ok[u]) v[u] = *(const vec_t*)(a.Fi + (int64_t)i * a.ldi + sub * VE);
                }
#pragma unroll
                for (int u = 0; u < SRP_U; ++u) {
                    if (s0 + u * EPW >= nb) break;                       // (wave-uniform)
                    const int ent = s0 + u * EPW + grp;
                    TM g[VE];
#pragma unroll
                    for (int t = 0; t < VE; ++t) g[t] = GatherT<TG>::get(v[u][t]);
                    TM e = __shfl(my_e, ent & 63, 64);
                    if (a.mode != SRP_APPLY) {
                        const TM r = __shfl(my_r, ent & 63, 64);
                        TM x = (TM)0;
#pragma unroll
                        for (int t = 0; t < VE; ++t) x += fo[t] * g[t];
#pragma unroll
                        for (int off = GL / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
                        if (a.mode == SRP_ERR) {
                            if (sub == 0 && ok[u]) sq += (double)((r - x) * (r - x)) - (double)((r - e - x) * (r - e - x));
                            continue;
                        }
                        e = ok[u] ? r - x : (TM)0;
                        if (a.evals && sub == 0 && ok[u]) a.evals[q0 + ent] = e;
                    }
                    if (!ok[u]) e = (TM)0;
#pragma unroll
                    for (int t = 0; t < VE; ++t) acc[t] += e * g[t];
                }
            }
        }
        if (a.mode != SRP_ERR) {
#pragma unroll
            for (int off = GL; off < 64; off <<= 1)             // the EPW lane groups of the wave, fixed order
#pragma unroll
                for (int t = 0; t < VE; ++t) acc[t] += __shfl_xor(acc[t], off, 64);
            if (grp == 0) {
                TM* dst = a.out + (int64_t)part * a.part_stride + o * a.ld_out + sub * VE;
#pragma unroll
                for (int t = 0; t < VE; ++t) dst[t] = acc[t];
            }
        }
    }
    if (a.mode == SRP_ERR) {
        sq = wave_sum(sq);
        if (lane == 0) a.sq[(int64_t)blockIdx.x * 4 + wv] = sq;
    }
}

// bf16 vectors of width w = 8 * GL (the bf16 engine's hot shapes: ranks 64 / 128 / 256), tuned form of srp_vec_kernel.
// Measured on config 5's ratings relation (80 M known entries, 256-byte rows, profiles/r03_srp_probe.txt): the pass
// WITHOUT dot products gathers 17.5 TB/s once 4 rows per lane group are in flight (7.5 TB/s with 2: the gathers are
// latency-bound L2 / Infinity Cache hits, pinning the parts to XCDs changes nothing), the first version WITH them only
// 10 TB/s -- ds_bpermute reductions and VALU conversions, not memory.  Hence here:
//   * 16 adjacent lanes read ONE contiguous row (full 128-byte lines per request; a layout that feeds the rows straight
//     into v_mfma_f32_16x16x32_bf16 as the B operand -- lane l = entry l & 15, chunk l >> 4 -- needs no reduction at all
//     but touches 16 half-used lines per load and gathered 9 TB/s even without the dot products);
//   * dot products as v_dot2c_f32_bf16 on the packed pairs (4 instead of 16 VALU operations per 8 elements), summed over
//     the lane group by DPP adds (quad_perm / row_half_mirror / row_mirror: vector-ALU rate, no LDS crossbar);
//   * indices / values are loaded a batch at a time (one coalesced load) and handed to the lane groups by DPP row
//     broadcasts (loading them per lane group instead -- one broadcast address per group, a loop trip ahead -- costs a
//     vector-memory instruction each and was slower: 12.6 TB/s without dot products; ds_bpermute hand-outs sit on the
//     LDS pipe), the residuals go back with one coalesced store per batch;
//   * the mode is a template parameter and out-of-range slots are clamped, not branched around.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// x + <a, b> over the eight bf16 of two 16-byte chunks: four v_dot2c_f32_bf16.  (The dwords are copied out of the vectors
// first: host clang's __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index -- seen in the emulator
// build; hipcc is not affected.)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float x) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), x, false);
}
__device__ __forceinline__ float dot8_bf16(u32x4 a, u32x4 b, float x) {
    const uint32_t a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
    return dot2_bf16(a3, b3, dot2_bf16(a2, b2, dot2_bf16(a1, b1, dot2_bf16(a0, b0, x))));
}

// lane k of every row of 16 lanes, broadcast to the row (DPP row_newbcast: vector-ALU rate, no LDS crossbar)
template <int K>
__device__ __forceinline__ int row_bcast(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x150 + K, 0xF, 0xF, true); }
template <int K>
__device__ __forceinline__ float row_bcast(float x) { return __builtin_bit_cast(float, row_bcast<K>(__builtin_bit_cast(int, x))); }

// GL lanes per gathered vector (w = 8 GL: GL = 8, 16, 32), MODE = SRP_*.  A batch is the entries one coalesced load of
// the index / value lists hands to the wave -- every ROW of 16 lanes holds the 16 entries its own lane group(s) will
// process, entry k of a row being broadcast to it by DPP (GL = 16: 64 entries, row g = group g takes 16 g + k;
// GL = 8: the two groups of a row take its entries k and 8 + k; GL = 32: 32 entries, both rows of a group hold its 16).
// Out-of-range slots of the last batch gather a valid row and contribute e = 0: no divergent branches in the loop.
template <int GL, int MODE>
__global__ __launch_bounds__(256) void srp_bf16_kernel(SrpArgs<uint16_t, float> a) {
    constexpr int EPB = GL == 32 ? 32 : 64;     // entries per batch
    constexpr int STEPS = GL == 8 ? 8 : 16;     // wave steps per batch
    constexpr int NU = 4;                       // wave steps (gathers per lane) in flight
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane / GL, sub = lane % GL;
    // the entry of the batch this lane loads, and the first entry of its group
    const int slot = GL == 32 ? (lane >> 5) * 16 + (lane & 15) : lane;
    const int gbase = GL == 32 ? (lane >> 5) * 16 : GL == 16 ? (lane >> 4) * 16 : (lane >> 3) * 8;
    const bool upper = GL == 8 && (lane & 8) != 0;          // GL = 8: the group that takes the upper half of its row
    int part;
    int64_t first;
    srp_segment(a.parts, part, first);
    const int64_t o = first + wv;
    double sq = 0.0;
    if (o < a.n_out) {                                  // (wave-uniform)
        const int64_t qa = a.ptr[o * a.parts + part], qb = a.ptr[o * a.parts + part + 1];
        u32x4 fo = {0u, 0u, 0u, 0u};
        if (MODE != SRP_APPLY) fo = *(const u32x4*)(a.Fo + o * a.ldo + sub * 8);
        float acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = 0.f;
        for (int64_t q0 = qa; q0 < qb; q0 += EPB) {
            const int nb = (int)(qb - q0 < EPB ? qb - q0 : EPB);         // entries of this batch
            const int64_t qm = q0 + (slot < nb ? slot : nb - 1);         // (clamped: always a valid entry)
            const int my_i = a.idx[qm];
            const float my_r = MODE != SRP_APPLY ? a.rvals[qm] : 0.f;
            const float my_e = MODE != SRP_RESIDUAL ? a.evals[qm] : 0.f;
            float e_out = 0.f;                                           // SRP_RESIDUAL: the residual of this lane's entry
            auto chunk = [&](auto c0) {                                  // NU steps: k = c0 .. c0 + NU - 1
                constexpr int K0 = decltype(c0)::value;
                u32x4 v[NU];
                float r[NU], e[NU];
                auto fetch = [&](auto uu) {
                    constexpr int U = decltype(uu)::value, K = K0 + U;
                    int i = row_bcast<K>(my_i);
                    r[U] = MODE != SRP_APPLY ? row_bcast<K>(my_r) : 0.f;
                    e[U] = MODE != SRP_RESIDUAL ? row_bcast<K>(my_e) : 0.f;
                    if (GL == 8) {                                       // the upper group of a row: entry 8 + k
                        const int i2 = row_bcast<K + 8>(my_i);
                        i = upper ? i2 : i;
                        if (MODE != SRP_APPLY) { const float r2 = row_bcast<K + 8>(my_r); r[U] = upper ? r2 : r[U]; }
                        if (MODE != SRP_RESIDUAL) { const float e2 = row_bcast<K + 8>(my_e); e[U] = upper ? e2 : e[U]; }
                    }
                    v[U] = *(const u32x4*)(a.Fi + (int64_t)i * a.ldi + sub * 8);
                };
                fetch(std::integral_constant<int, 0>());
                fetch(std::integral_constant<int, 1>());
                fetch(std::integral_constant<int, 2>());
                fetch(std::integral_constant<int, 3>());
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const bool ok = gbase + K0 + u < nb;
                    float ev = e[u];
                    if (MODE != SRP_APPLY) {
                        float x = dot8_bf16(fo, v[u], 0.f);
                        x = dpp_add<0xB1>(x);                            // quad_perm [1,0,3,2]
                        x = dpp_add<0x4E>(x);                            // quad_perm [2,3,0,1]
                        x = dpp_add<0x141>(x);                           // row_half_mirror
                        if (GL >= 16) x = dpp_add<0x140>(x);             // row_mirror
                        if (GL >= 32) x += __shfl_xor(x, 16, 64);
                        if (MODE == SRP_ERR) {
                            if (sub == 0 && ok) sq += (double)((r[u] - x) * (r[u] - x)) - (double)((r[u] - ev - x) * (r[u] - ev - x));
                            continue;
                        }
                        ev = r[u] - x;
                        if ((lane & (GL == 8 ? 7 : 15)) == K0 + u) e_out = ev;      // the lane that loaded this entry keeps it
                    }
                    if (!ok) ev = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {                        // two bf16 per dword: low half, high half
                        union { uint32_t u; float f; } lo, hi;
                        lo.u = v[u][t] << 16;
                        hi.u = v[u][t] & 0xffff0000u;
                        acc[2 * t] += ev * lo.f;
                        acc[2 * t + 1] += ev * hi.f;
                    }
                }
            };
            const int steps = nb < STEPS ? nb : STEPS;                   // (a group runs out of entries when the batch does)
            chunk(std::integral_constant<int, 0>());
            if (steps > 4) chunk(std::integral_constant<int, 4>());
            if (STEPS > 8) {
                if (steps > 8) chunk(std::integral_constant<int, 8>());
                if (steps > 12) chunk(std::integral_constant<int, 12>());
            }
            // one coalesced store of the batch's residuals (GL = 32: the first row of every group)
            if (MODE == SRP_RESIDUAL && a.evals && slot < nb && (GL != 32 || (lane & 16) == 0)) a.evals[q0 + slot] = e_out;
        }
        if (MODE != SRP_ERR) {
#pragma unroll
            for (int off = GL; off < 64; off <<= 1)                     // the lane groups of the wave, fixed order
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] += __shfl_xor(acc[t], off, 64);
            if (grp == 0) {
                float* dst = a.out + (int64_t)part * a.part_stride + o * a.ld_out + sub * 8;
#pragma unroll
                for (int t = 0; t < 8; ++t) dst[t] = acc[t];
            }
        }
    }
    if (MODE == SRP_ERR) {
        sq = wave_sum(sq);
        if (lane == 0) a.sq[(int64_t)blockIdx.x * 4 + wv] = sq;
    }
}

// srp_bf16_v6_kernel: the tuned pass with the vector-ALU work per entry almost halved (round 3, second half).  The ISA
// of srp_bf16_kernel shows why its dot-product flavours ran at 11-13 TB/s of gathered rows against 17.5 TB/s for the
// bare gathers: 49 vector-ALU instructions per wave step (4 entries), the pass is issue-bound -- 64-bit address
// arithmetic (two quarter-rate v_mul_lo_u32 and a v_mad_u64_u32 per row), multiplies and adds of the accumulation left
// unfused behind the shuffles of the SLP pass, a compare and two selects per step for the slots past the end of a list.
// Here, per step (CPL = 1): one v_add_u32_dpp (the lane that loaded entry k of the row holds its BYTE offset, 32 bits:
// the gathered matrix is below 4 GiB, checked at launch), the 16-byte load off a scalar base, four v_dot2c, the
// butterfly as four v_add_f32_dpp, one v_sub_f32_dpp (r of lane k minus x), one select to keep the residual, eight
// conversions and four v_pk_fma_f32 (explicit two-element vectors; the compiler emits the DPP forms itself once the
// accumulation is out of the SLP pass's way): 25.  Slots past the end of a list point at an all-zero row kept behind the gathered matrix
// (SrpArgs::zero_off) with r = 0 (e = 0 in SRP_APPLY): they contribute exactly nothing without a mask in the loop.
// A row of 16 lanes always covers one gathered vector: w = 128 with one 16-byte chunk per lane (CPL = 1), w = 256 with
// two (CPL = 2: chunks l and 16 + l, two coalesced 256-byte halves) -- the butterfly stays inside the row; w = 64 with one
// 8-byte chunk per lane (DW = 2; the 0/1 relations kept as lists, SRP_ONES: every entry counts 1, no value list at all).
// The list bounds are moved to scalar registers: the loop and its early exits branch on scalar conditions.
// (A first version issued the DPP forms as inline assembly: the hazard recogniser does not look into assembly -- DPP
// after an EXEC write, any other vector-ALU read of a v_dot2c result -- and the s_nop padding that looked sufficient was
// not under low occupancy: wrong, run-to-run different sums in the eight-loads-in-flight variants.  Builtins only.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int64_t wave_uniform(int64_t x) {               // a value every lane holds, to scalar registers
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)x >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
enum { SRP_ONES = 3 };      // srp_bf16_v6_kernel only: every entry counts 1 (0/1 relations as lists: out[o] = sum of Fi[i])
template <int DW> struct SrpChunk;                                         // DW dwords (2 DW bf16) of a gathered row per lane
template <> struct SrpChunk<4> { typedef u32x4 type; };
template <> struct SrpChunk<2> { typedef unsigned int type __attribute__((ext_vector_type(2))); };
// CPL chunks of DW dwords per lane: w = 16 lanes x CPL x 2 DW elements (DW = 4: 128 / 256, DW = 2: 64 / 128)
template <int CPL, int MODE, int DW = 4>
__global__ __launch_bounds__(256) void srp_bf16_v6_kernel(SrpArgs<uint16_t, float> a) {
    static_assert(CPL == 1 || CPL == 2, "one or two chunks per lane");
    static_assert(DW == 4 || DW == 2, "16- or 8-byte chunks");
    static_assert(MODE == SRP_RESIDUAL || MODE == SRP_APPLY || MODE == SRP_ONES, "the error pass stays on srp_bf16_kernel");
    typedef typename SrpChunk<DW>::type chunk_t;
    constexpr int NU = 4;                       // wave steps (gathered rows per lane group) in flight
    constexpr int CE = 2 * DW;                  // elements of a chunk per lane; a chunk spans 16 CE elements of the row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    int part;
    int64_t first;
    srp_segment(a.parts, part, first);
    const int64_t o = first + wv;
    if (o >= a.n_out) return;                                              // (wave-uniform)
    const int64_t qa = wave_uniform(a.ptr[o * a.parts + part]), qb = wave_uniform(a.ptr[o * a.parts + part + 1]);
    uint32_t fo[CPL][DW];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
#pragma unroll
        for (int t = 0; t < DW; ++t) fo[c][t] = 0u;
        if (MODE == SRP_RESIDUAL) {
            const chunk_t v = *(const chunk_t*)(a.Fo + o * a.ldo + c * 16 * CE + sub * CE);
#pragma unroll
            for (int t = 0; t < DW; ++t) fo[c][t] = v[t];
        }
    }
    f32x2 acc[DW * CPL];
#pragma unroll
    for (int t = 0; t < DW * CPL; ++t) acc[t] = f32x2{0.f, 0.f};
    const unsigned char* fi = (const unsigned char*)a.Fi;
    const uint32_t ldb = (uint32_t)a.ldi * 2u, subb = (uint32_t)sub * (4u * DW);
    for (int64_t q0 = qa; q0 < qb; q0 += 64) {
        const int nb = (int)(qb - q0 < 64 ? qb - q0 : 64);                 // entries of this batch: lane l loads entry l
        const bool mine = lane < nb;
        const int64_t qm = q0 + (mine ? lane : nb - 1);                    // (clamped: always a valid address)
        const uint32_t ld_i = (uint32_t)a.idx[qm];
        const float ld_v = MODE == SRP_RESIDUAL ? a.rvals[qm] : MODE == SRP_APPLY ? a.evals[qm] : 1.f;
        const int my_off = (int)(mine ? ld_i * ldb : a.zero_off);
        const float my_v = mine ? ld_v : 0.f;                              // r of the entry (SRP_APPLY: its stored residual)
        float e_out = 0.f;                                                 // SRP_RESIDUAL: the residual of this lane's entry
        auto chunk = [&](auto c0) {                                        // NU steps: k = c0 .. c0 + NU - 1
            constexpr int K0 = decltype(c0)::value;
            uint32_t v[NU][CPL][DW];
            static_for<NU>([&](auto uu) {
                constexpr int U = decltype(uu)::value;
                const unsigned char* src = fi + ((uint32_t)row_bcast<K0 + U>(my_off) + subb);
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const chunk_t x = *(const chunk_t*)(src + c * 64 * DW);
#pragma unroll
                    for (int t = 0; t < DW; ++t) v[U][c][t] = x[t];
                }
            });
            static_for<NU>([&](auto uu) {
                constexpr int U = decltype(uu)::value, K = K0 + U;
                float ev;
                if (MODE == SRP_RESIDUAL) {
                    float x = 0.f;
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
#pragma unroll
                        for (int t = 0; t < DW; ++t) x = dot2_bf16(fo[c][t], v[U][c][t], x);
                    x = dpp_add<0xB1>(x);                                  // quad_perm [1,0,3,2]
                    x = dpp_add<0x4E>(x);                                  // quad_perm [2,3,0,1]
                    x = dpp_add<0x141>(x);                                 // row_half_mirror
                    x = dpp_add<0x140>(x);                                 // row_mirror
                    ev = row_bcast<K>(my_v) - x;
                    e_out = sub == K ? ev : e_out;                         // the lane that loaded this entry keeps it
                } else {
                    ev = row_bcast<K>(my_v);
                }
                const f32x2 e2 = {ev, ev};
#pragma unroll
                for (int c = 0; c < CPL; ++c)
#pragma unroll
                    for (int t = 0; t < DW; ++t) {                         // two bf16 per dword: low half, high half
                        const uint32_t w = v[U][c][t];
                        const f32x2 g = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
                        acc[DW * c + t] = __builtin_elementwise_fma(g, e2, acc[DW * c + t]);
                    }
            });
        };
        chunk(std::integral_constant<int, 0>());                          // (a group runs out of entries when the batch does)
        if (nb > 4) chunk(std::integral_constant<int, 4>());
        if (nb > 8) chunk(std::integral_constant<int, 8>());
        if (nb > 12) chunk(std::integral_constant<int, 12>());
        if (MODE == SRP_RESIDUAL && a.evals && mine) a.evals[q0 + lane] = e_out;      // one coalesced store per batch
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        float out[CE];
#pragma unroll
        for (int t = 0; t < DW; ++t) { out[2 * t] = acc[DW * c + t].x; out[2 * t + 1] = acc[DW * c + t].y; }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1)                            // the lane groups of the wave, fixed order
#pragma unroll
            for (int t = 0; t < CE; ++t) out[t] += __shfl_xor(out[t], off, 64);
        if (grp == 0) {
            float* dst = a.out + (int64_t)part * a.part_stride + o * a.ld_out + c * 16 * CE + sub * CE;
#pragma unroll
            for (int t = 0; t < CE; ++t) dst[t] = out[t];
        }
    }
}

// segment pointers of lists cut into parts of the inner index (ascending within a list): seg[o * parts + p] = the first
// entry of list o whose inner index is >= p * pw (binary search; one thread per segment), seg[n_out * parts] = the end
static __global__ __launch_bounds__(256) void parted_ptr_kernel(const int64_t* __restrict__ ptr, const int* __restrict__ idx, int64_t n_out,
                                                         int parts, int64_t pw, int64_t* __restrict__ seg) {
    const int64_t total = n_out * parts;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= total; e += (int64_t)gridDim.x * blockDim.x) {
        if (e == total) { seg[e] = ptr[n_out]; continue; }
        const int64_t o = e / parts, lim = (e % parts) * pw;
        int64_t lo = ptr[o], hi = ptr[o + 1];
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (idx[mid] < lim) lo = mid + 1; else hi = mid;
        }
        seg[e] = lo;
    }
}

// any width w <= 64 * SRP_MAXREP: lane l holds the elements l, l + 64, ... of the vectors; one entry per step
constexpr int SRP_MAXREP = 16;
template <typename TG, typename TM>
__global__ __launch_bounds__(256) void srp_any_kernel(SrpArgs<TG, TM> a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int part;
    int64_t first;
    srp_segment(a.parts, part, first);
    const int64_t o = first + wv;
    double sq = 0.0;
    if (o < a.n_out) {
        const int64_t qa = a.ptr[o * a.parts + part], qb = a.ptr[o * a.parts + part + 1];
        const int reps = (a.w + 63) / 64;
        TM fo[SRP_MAXREP], acc[SRP_MAXREP];
#pragma unroll
        for (int t = 0; t < SRP_MAXREP; ++t) {
            const int j = lane + 64 * t;
            fo[t] = (a.mode != SRP_APPLY && t < reps && j < a.w) ? GatherT<TG>::get(a.Fo[o * a.ldo + j]) : (TM)0;
            acc[t] = (TM)0;
        }
        for (int64_t q = qa; q < qb; ++q) {
            const int64_t i = a.idx[q];
            TM g[SRP_MAXREP];
#pragma unroll
            for (int t = 0; t < SRP_MAXREP; ++t) {
                const int j = lane + 64 * t;
                g[t] = (t < reps && j < a.w) ? GatherT<TG>::get(a.Fi[i * a.ldi + j]) : (TM)0;
            }
            TM e = (TM)0;
            if (a.mode != SRP_RESIDUAL) e = a.evals[q];
            if (a.mode != SRP_APPLY) {
                TM x = (TM)0;
#pragma unroll
                for (int t = 0; t < SRP_MAXREP; ++t) x += fo[t] * g[t];
                x = wave_sum(x);
                const TM r = a.rvals[q];
                if (a.mode == SRP_ERR) {
                    if (lane == 0) sq += (double)((r - x) * (r - x)) - (double)((r - e - x) * (r - e - x));
                    continue;
                }
                e = r - x;
                if (a.evals && lane == 0) a.evals[q] = e;
            }
#pragma unroll
            for (int t = 0; t < SRP_MAXREP; ++t) acc[t] += e * g[t];
        }
        if (a.mode != SRP_ERR) {
            TM* dst = a.out + (int64_t)part * a.part_stride + o * a.ld_out;
#pragma unroll
            for (int t = 0; t < SRP_MAXREP; ++t) {
                const int j = lane + 64 * t;
                if (t < reps && j < a.w) dst[j] = acc[t];
            }
        }
    }
    if (a.mode == SRP_ERR) {
        sq = wave_sum(sq);
        if (lane == 0) a.sq[(int64_t)blockIdx.x * 4 + wv] = sq;
    }
}

// dst[e] = part 0 + part 1 + ... (fixed order).  16-byte chunks where the layout allows it (the strides of the passes are
// multiples of the rank: always, for the 16-byte-chunk list kernels): the parts of a chunk are independent loads in flight
// (8 parts of config 5's 40k x 128 column side: 164 MB, 89 -> us with one element per thread and load)
template <typename T>
__global__ __launch_bounds__(256) void sum_parts_kernel(T* __restrict__ dst, const T* __restrict__ parts, int64_t stride,
                                                        int nparts, int64_t total) {
    constexpr int VE = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    const bool vec = total % VE == 0 && stride % VE == 0 && (((uintptr_t)dst | (uintptr_t)parts) & 15) == 0;
    if (vec) {
        const int64_t nv = total / VE, sv = stride / VE;
        const vec_t* src = (const vec_t*)parts;
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nv; e += (int64_t)gridDim.x * blockDim.x) {
            vec_t s = src[e];
            for (int p = 1; p < nparts; ++p) s += src[(int64_t)p * sv + e];
            ((vec_t*)dst)[e] = s;
        }
        return;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        T s = parts[e];
        for (int p = 1; p < nparts; ++p) s += parts[(int64_t)p * stride + e];
        dst[e] = s;
    }
}

// E (+)= max(A, 0) ; D (+)= max(-A, 0)      (the +- split of _dfmc.py:141-144 on an already formed product)
template <typename T>
__global__ __launch_bounds__(256) void split_accumulate_kernel(T* __restrict__ E, T* __restrict__ D, const T* __restrict__ A,
                                                               int64_t total, int accumulate) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const T v = A[e];
        const T p = v > (T)0 ? v : (T)0, n = v > (T)0 ? (T)0 : -v;
        if (accumulate) {
            E[e] += p;
            D[e] += n;
        } else {
            E[e] = p;
            D[e] = n;
        }
    }
}

// out[0] = sum_e A[e] * B[e] (f64, one workgroup; c x c operands)
static __global__ __launch_bounds__(256) void dot_small_kernel(const double* __restrict__ A, const double* __restrict__ B, int64_t total,
                                                        double scale, double* __restrict__ out, int accumulate) {
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t e = threadIdx.x; e < total; e += 256) s += A[e] * B[e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = scale * (red[0] + red[1] + red[2] + red[3]);
        out[0] = accumulate ? out[0] + t : t;
    }
}

// ---- bind time: the known entries of a masked relation from its packed mask (bit = 1: unknown) ----------------
// word w of mask row r with the columns >= cols cleared, inverted: the KNOWN entries of columns [64 w, 64 w + 64)
__device__ __forceinline__ unsigned long long known_word(const uint8_t* __restrict__ Mb, int64_t ldmb, int64_t r, int64_t w,
                                                         int64_t cols) {
    const unsigned long long m = *(const unsigned long long*)(Mb + r * ldmb + w * 8);
    const int64_t left = cols - w * 64;
    const unsigned long long valid = left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
    return ~m & valid;
}

// counts[r * parts + p] = known entries of row r in the columns [p * part_w, (p + 1) * part_w)   (part_w % 64 == 0)
static __global__ __launch_bounds__(256) void known_row_count_kernel(const uint8_t* __restrict__ Mb, int64_t ldmb, int64_t rows,
                                                              int64_t cols, int parts, int64_t part_w, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t wpp = part_w >> 6;
    for (int64_t r = wave; r < rows; r += nwaves)
        for (int p = 0; p < parts; ++p) {
            const int64_t w0 = p * wpp, w1 = (w0 + wpp) * 64 < cols ? w0 + wpp : (cols + 63) / 64;
            int cnt = 0;
            for (int64_t w = w0 + lane; w < w1; w += 64) cnt += __popcll(known_word(Mb, ldmb, r, w, cols));
            cnt = wave_sum(cnt);
            if (lane == 0) counts[r * parts + p] = cnt;
        }
}

// idx / rvals of row r from ptr[r * parts] on: the known columns, ascending, and the relation's values there
template <typename TR, typename TM>
__global__ __launch_bounds__(256) void known_row_fill_kernel(const uint8_t* __restrict__ Mb, int64_t ldmb, int64_t rows, int64_t cols,
                                                             int parts, const int64_t* __restrict__ ptr, const TR* __restrict__ R,
                                                             int64_t ldr, int* __restrict__ idx, TM* __restrict__ rvals) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t nw = (cols + 63) / 64;
    for (int64_t r = wave; r < rows; r += nwaves) {
        int64_t base = ptr[r * parts];
        for (int64_t k0 = 0; k0 < nw; k0 += 64) {
            unsigned long long x = (k0 + lane < nw) ? known_word(Mb, ldmb, r, k0 + lane, cols) : 0ull;
            const int pc = __popcll(x);
            const int incl = wave_incl_scan(pc);
            int64_t at = base + incl - pc;
            while (x) {
                const int b = __ffsll((long long)x) - 1;
                x &= x - 1ull;
                const int64_t c = (k0 + lane) * 64 + b;
                idx[at] = (int)c;
                rvals[at] = (TM)GatherT<TR>::get(R[r * ldr + c]);
                ++at;
            }
            base += __shfl(incl, 63, 64);
        }
    }
}

// transpose of the row lists, cut into row parts of part_h rows: per (column, part) counts / arbitrary-order fill
static __global__ __launch_bounds__(256) void known_col_count_kernel(const int64_t* __restrict__ rptr, const int* __restrict__ idx, int rparts,
                                                              int64_t rows, int cparts, int64_t part_h, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const int p = (int)(r / part_h);
        for (int64_t q = rptr[r * rparts] + lane; q < rptr[(r + 1) * rparts]; q += 64) atomicAdd(&counts[(int64_t)idx[q] * cparts + p], 1);
    }
}
static __global__ __launch_bounds__(256) void known_col_fill_kernel(const int64_t* __restrict__ rptr, const int* __restrict__ idx, int rparts,
                                                             int64_t rows, int cparts, int64_t part_h, const int64_t* __restrict__ cptr,
                                                             int* __restrict__ fillpos, int* __restrict__ cidx) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const int p = (int)(r / part_h);
        for (int64_t q = rptr[r * rparts] + lane; q < rptr[(r + 1) * rparts]; q += 64) {
            const int64_t seg = (int64_t)idx[q] * cparts + p;
            cidx[cptr[seg] + atomicAdd(&fillpos[seg], 1)] = (int)r;
        }
    }
}
// (every segment's rows are then sorted ascending by csc_sort_kernel over the cols * cparts segments)
template <typename TR, typename TM>
__global__ __launch_bounds__(256) void known_col_values_kernel(const int64_t* __restrict__ cptr, const int* __restrict__ cidx, int cparts,
                                                               int64_t cols, const TR* __restrict__ R, int64_t ldr,
                                                               TM* __restrict__ rvals) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave; c < cols; c += nwaves)
        for (int64_t q = cptr[c * cparts] + lane; q < cptr[(c + 1) * cparts]; q += 64)
            rvals[q] = (TM)GatherT<TR>::get(R[(int64_t)cidx[q] * ldr + c]);
}

}  // namespace skf
