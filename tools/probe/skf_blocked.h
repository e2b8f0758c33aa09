// skf_blocked.h -- the known-entry passes of DFMC as BLOCKED SDDMM + SpMM on the matrix cores (round 6).
// PROBE ONLY (tools/probe/blk_probe.hip): measured against srp_bf16_v6_kernel and NOT taken into the product -- the W pass
// reached 0.87 ms against 1.12 (the bar was 0.8), the residual passes stayed behind the per-entry kernels; numbers and the
// reasons in profiles/r06_blocked_lists_probe.txt (two forms: 16 waves x 1 strip, 8 waves x 2 strips).
//
// What skf_known.h computes entry by entry -- per known entry (o, i) of a masked relation one gathered 256-byte bf16 row
// of the inner factor, a dot product on the vector ALU and an axpy into the outer row's accumulator (reference
// _dfmc.py:319-325 + :152-170 restated on lists) -- is regrouped here so that the inner rows are read from memory ONCE
// per (256 outer rows x 256 inner rows) block instead of once per entry:
//   * outer objects in STRIPS of 16 (one wave, the strip's 16 x 128 f32 accumulators in registers), 16 strips = 256 outer
//     objects per workgroup (1024 threads);
//   * inner objects in BLOCKS of 256, staged as they lie (bf16 rows, 256 B + 32 B of padding = 288-byte pitch) by LDS-DMA
//     into one of two 72 KiB buffers while the other is being consumed: one barrier per block;
//   * the entries of a (strip, block) CELL in GROUPS of 16 (padded with null entries).  A group is one matrix-core step:
//       SDDMM   D'[entry][outer] = Fi[entry] . Fo[outer]        4 x v_mfma_f32_16x16x32_bf16 (K = the 128 columns);
//               A = the 16 gathered rows (ds_read_b128 from the staged block), B = the strip's 16 outer rows (registers)
//       pick    lane (outer o, lane group g) holds D' of the entries 4g .. 4g+3 against o: the entry whose outer object
//               IS o gives e = r - x there; every other lane's value is dropped (an entry has one outer object)
//       SpMM    out[outer][col] += sum_entries A[outer][entry] Fi[entry][col]     8 x v_mfma_f32_16x16x32_bf16, one per
//               16 columns; A = e on the entry's own outer row, 0 elsewhere, as bf16 hi + lo (the 32 K slots = 16 entries x
//               {hi, lo}: 16 mantissa bits of e), B = the same gathered rows read TRANSPOSED (ds_read_b64_tr_b16: lane
//               group g gets its 4 entries' values of the lane's column -- used for the hi and the lo slots alike).
//     The W pass (BLK_APPLY) has no SDDMM: e comes stored (packed hi | lo << 16 by the column pass of the iteration before).
// Per entry: 512 B of LDS reads and 12 matrix-core steps / 16 (BLK_RESIDUAL), 256 B and 8 / 16 (BLK_APPLY); the staged bytes
// are n_out / 256 x the inner matrix (config 5: 4 GB per pass from L2 against 20.5 GB of per-entry gathers).
//
// LDS banks: a transposed read touches 8 entries x 32 B per half wave, a ds_read_b128 pass 16 entries x 16 B.  With the
// 288-byte pitch both are conflict-free when the entry at position p of its group lies in a block row with
// (row & 7) == (p & 7); the lists are ARRANGED that way at bind time as far as the cell's rows allow (two positions per
// residue and group; what does not fit goes to the free positions and costs a conflict).
//
// Summation order: fixed by the lists (cells ascending, groups ascending, the matrix-core instruction's own order inside
// a group): results are run-to-run deterministic; parts of the inner blocks write partial outputs that are summed in a
// fixed order (sum_parts_kernel), exactly as in skf_known.h.
#pragma once
#include "skf_known.h"

namespace skf {

constexpr int BLK_OS = 16;                        // outer objects per strip (one wave)
constexpr int BLK_WAVES = 16;                     // strips per workgroup
constexpr int BLK_WGR = BLK_OS * BLK_WAVES;       // outer objects per workgroup
constexpr int BLK_IB = 256;                       // inner objects per staged block
constexpr int BLK_W = 128;                        // width of the vectors (bf16 elements)
constexpr int BLK_PITCH = 288;                    // bytes per staged row
constexpr int BLK_BUF = BLK_IB * BLK_PITCH;       // one staged block
constexpr int BLK_LDS = 2 * BLK_BUF;              // 147 456 B
constexpr int BLK_DMA = BLK_BUF / 1024;           // wave-wide LDS-DMA instructions per block (72)
constexpr int BLK_NULL = 0xFF;                    // outer index of a padding entry: matches no lane

enum { BLK_APPLY = 0, BLK_RESIDUAL = 1 };

struct BlkArgs {
    const int* cellptr;        // [strips][nblk + 1]: first GROUP of cell (strip, block); a strip's groups are contiguous
    const uint32_t* meta;      // [groups][8]: inner local index of the 16 entries (u8 each), then their outer local index
    const float* rvals;        // [groups][16]: the relation at the entry (BLK_RESIDUAL)
    uint32_t* evals;           // [groups][16]: residuals as bf16 hi | lo << 16 (BLK_APPLY reads; BLK_RESIDUAL writes when given)
    const uint16_t* Fo;        // [n_out][ldo]  vectors of the outer objects (BLK_RESIDUAL)
    const uint16_t* Fi;        // [n_in][ldi]   vectors of the inner objects (staged)
    float* out;                // [parts][n_out][ld_out]
    int64_t ldo, ldi, ld_out, part_stride, n_out, n_in;
    int nblk;                  // inner blocks
    int parts;                 // 1, 2, 4 or 8 parts of the inner blocks (pinned to XCDs as in skf_known.h)
    int blk_per_part;
};

__device__ __host__ __forceinline__ uint32_t blk_pack_hi_lo(float e) {
    const uint32_t hi = f32_to_bf16_rne(e);
    const uint32_t lo = f32_to_bf16_rne(e - bf16_to_f32((uint16_t)hi));
    return hi | (lo << 16);
}
__device__ __host__ __forceinline__ float blk_unpack_hi_lo(uint32_t w) {
    return bf16_to_f32((uint16_t)(w & 0xFFFFu)) + bf16_to_f32((uint16_t)(w >> 16));
}

template <int K>
__device__ __forceinline__ uint32_t row_bcast_u(uint32_t x) { return (uint32_t)row_bcast<K>((int)x); }
typedef __bf16 blk_bf16x2 __attribute__((ext_vector_type(2)));
// two f32 -> two bf16 (round to nearest even) in one dword: v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t blk_cvt_pk(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, blk_bf16x2));
}

// VAR (bound-finding builds of tools/probe/blk_probe.hip only; the product instantiates VAR = 0): 1 = no matrix-core steps,
// 2 = no per-group vector-ALU preparation (a constant operand A), 3 = no LDS reads (constant operands B)
template <int MODE, int VAR = 0>
__global__ __launch_bounds__(1024) void blk_pass_kernel(BlkArgs a) {
    HIP_DYNAMIC_SHARED(unsigned char, lds)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);               // (scalar: everything derived from it stays scalar)
    const int sub = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7;
    const int lg = a.parts >= 8 ? 3 : a.parts >= 4 ? 2 : a.parts >= 2 ? 1 : 0;
    const int part = xcd & (a.parts - 1);
    const int64_t ob = (int64_t)(blockIdx.x >> 3) * (8 >> lg) + (xcd >> lg);
    if (ob * BLK_WGR >= a.n_out) return;                                   // (workgroup-uniform)
    const int b0 = part * a.blk_per_part;
    const int b1 = b0 + a.blk_per_part < a.nblk ? b0 + a.blk_per_part : a.nblk;
    const int64_t strip = ob * BLK_WAVES + wv;
    const bool live = strip * BLK_OS < a.n_out;                            // (wave-uniform)
    const int* cp = a.cellptr + (live ? strip : 0) * (int64_t)(a.nblk + 1);

    // LDS-DMA of block `blk` into buffer `buf`: wave wv issues the 1 KiB pieces wv, wv + 16, ...; piece q covers the 16-byte
    // slots 64 q .. 64 q + 63 of the padded image (slot s = row s / 18, chunk s % 18; chunks 16, 17 are the padding)
    const unsigned char* fi = (const unsigned char*)a.Fi;
    const int64_t ldb = a.ldi * 2;
    auto dma_block = [&](int blk, int buf) {
        const int64_t base = (int64_t)blk * BLK_IB;
        const unsigned char* src = fi + base * ldb;                        // (wave-uniform: a scalar base, 32-bit lane offsets)
        const int64_t left = a.n_in - base;
        const int rmax = (int)(left < BLK_IB ? left : BLK_IB) - 1;         // rows past the end: never referenced by an entry
#pragma unroll
        for (int i = 0; i < (BLK_DMA + BLK_WAVES - 1) / BLK_WAVES; ++i) {
            const int q = wv + BLK_WAVES * i;
            if (q < BLK_DMA) {                                             // (wave-uniform)
                const int slot = q * 64 + lane;
                int row = (slot * 3641) >> 16;                             // slot / 18 for slot < 4608
                int w = slot - row * 18;
                w = w < 16 ? w : 15;
                row = row < rmax ? row : rmax;
                const uint32_t off = (uint32_t)row * (uint32_t)ldb + (uint32_t)w * 16u;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                                 (__attribute__((address_space(3))) void*)(lds + buf * BLK_BUF + q * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 fo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) fo[k] = u32x4{0u, 0u, 0u, 0u};
    if (MODE == BLK_RESIDUAL && live) {
        int64_t o = strip * BLK_OS + sub;
        o = o < a.n_out ? o : a.n_out - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) fo[k] = *(const u32x4*)(a.Fo + o * a.ldo + 32 * k + 8 * g);
    }

    // the metadata of a cell: lane (sub, g) loads, for GROUP first + sub of the cell, what lane group g needs of it -- the
    // inner and outer indices of its four entries and their values -- and a group's turn hands them to the row of 16 lanes
    // by DPP row broadcasts (no LDS, no reload per group)
    struct Meta {
        uint32_t il;         // the four inner indices of lane group g
        uint32_t ol;         // ... and their outer indices
        u32x4 v;
    };
    auto load_meta = [&](int first, int count) {
        Meta m;
        m.il = 0u;
        m.ol = 0xFFFFFFFFu;
        m.v = u32x4{0u, 0u, 0u, 0u};
        if (count > 0) {
            const int64_t gi = first + (sub < count ? sub : count - 1);
            const uint32_t* mp = a.meta + gi * 8;
            m.il = mp[g];
            m.ol = mp[4 + g];
            m.v = MODE == BLK_APPLY ? *(const u32x4*)(a.evals + gi * 16 + 4 * g) : *(const u32x4*)(a.rvals + gi * 16 + 4 * g);
        }
        return m;
    };

    // ---- one group of 16 entries (K-th of the batch whose metadata `mt` holds) against the staged block `buf`, in three
    // pieces so that a batch can run them software-pipelined: prep(K + 1) and the first reads of group K + 1 go out under
    // the matrix-core steps of group K (the LDS latency and the vector-ALU work of a group hide behind its predecessor)
    struct Prep {
        u32x4 af;                    // SpMM operand A of the group
        const unsigned char* rowb;   // this lane's address for the transposed reads
    };
    // the K-th group's share of a batch's metadata, handed to the rows of 16 lanes by DPP row broadcasts; the lane is an
    // instruction immediate, so a run-time K goes through a scalar switch (everything behind it is one rolled copy of code)
    struct Bc {
        uint32_t ol4, il4, v0, v1, v2, v3;
    };
    auto bcast = [&](int k, const Meta& mt) {
        Bc b;
        auto take = [&](auto kk) {
            constexpr int K = decltype(kk)::value;
            b.ol4 = row_bcast_u<K>(mt.ol); b.il4 = row_bcast_u<K>(mt.il);
            b.v0 = row_bcast_u<K>(mt.v.x); b.v1 = row_bcast_u<K>(mt.v.y); b.v2 = row_bcast_u<K>(mt.v.z); b.v3 = row_bcast_u<K>(mt.v.w);
        };
        switch (k) {
        case 0: take(std::integral_constant<int, 0>()); break;
        case 1: take(std::integral_constant<int, 1>()); break;
        case 2: take(std::integral_constant<int, 2>()); break;
        case 3: take(std::integral_constant<int, 3>()); break;
        case 4: take(std::integral_constant<int, 4>()); break;
        case 5: take(std::integral_constant<int, 5>()); break;
        case 6: take(std::integral_constant<int, 6>()); break;
        case 7: take(std::integral_constant<int, 7>()); break;
        case 8: take(std::integral_constant<int, 8>()); break;
        case 9: take(std::integral_constant<int, 9>()); break;
        case 10: take(std::integral_constant<int, 10>()); break;
        case 11: take(std::integral_constant<int, 11>()); break;
        case 12: take(std::integral_constant<int, 12>()); break;
        case 13: take(std::integral_constant<int, 13>()); break;
        case 14: take(std::integral_constant<int, 14>()); break;
        default: take(std::integral_constant<int, 15>()); break;
        }
        return b;
    };
    auto prep = [&](int k, const Meta& mt, int gfirst, const unsigned char* buf, bool exists = true) {
        Prep pr;
        const Bc bc = bcast(k, mt);
        const uint32_t ol4 = bc.ol4, il4 = bc.il4;
        const uint32_t v4[4] = {bc.v0, bc.v1, bc.v2, bc.v3};
        bool mine[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[j] = ((ol4 >> (8 * j)) & 0xFFu) == (uint32_t)sub;
        // SpMM operand A[outer = sub][K slot (g, j)] = hi (j < 4) / lo (j >= 4) half of e of entry 4g + (j & 3) where that
        // entry is sub's own, 0 elsewhere
        if (MODE == BLK_APPLY) {                       // e stored as hi | lo << 16
            uint32_t mk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mk[j] = mine[j] ? v4[j] : 0u;
            pr.af.x = __builtin_amdgcn_perm(mk[1], mk[0], 0x05040100u);
            pr.af.y = __builtin_amdgcn_perm(mk[3], mk[2], 0x05040100u);
            pr.af.z = __builtin_amdgcn_perm(mk[1], mk[0], 0x07060302u);
            pr.af.w = __builtin_amdgcn_perm(mk[3], mk[2], 0x07060302u);
        } else {
            // the row of entry `sub`: its index sits with lane group sub >> 2 -- the four groups' words through scalar registers
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)il4, 0), s1 = (uint32_t)__builtin_amdgcn_readlane((int)il4, 16),
                           s2 = (uint32_t)__builtin_amdgcn_readlane((int)il4, 32), s3 = (uint32_t)__builtin_amdgcn_readlane((int)il4, 48);
            const int wa = sub >> 2;
            const uint32_t ila4 = wa == 0 ? s0 : wa == 1 ? s1 : wa == 2 ? s2 : s3;
            const uint32_t ila = (ila4 >> (8 * (sub & 3))) & 0xFFu;
            // SDDMM: A = the 16 gathered rows (lane (m = sub, g): columns 32 k + 8 g .. + 7 of entry sub's row)
            const unsigned char* rowa = buf + ila * BLK_PITCH + g * 16;
            f32x4 dd = {0.f, 0.f, 0.f, 0.f};
            u32x4 a0 = lds_read_b128<0>(rowa), a1 = lds_read_b128<64>(rowa);
            lds_wait<1>(a0);
            dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, fo[0]), dd, 0, 0, 0);
            u32x4 a2 = lds_read_b128<128>(rowa);
            lds_wait<1>(a1);
            dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, fo[1]), dd, 0, 0, 0);
            u32x4 a3 = lds_read_b128<192>(rowa);
            lds_wait<1>(a2);
            dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, fo[2]), dd, 0, 0, 0);
            lds_wait<0>(a3);
            dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3), __builtin_bit_cast(bf16x8, fo[3]), dd, 0, 0, 0);
            // lane (o = sub, g), element j: <Fi[entry 4g + j], Fo[o]>; the entry's own outer object keeps e = r - x, split
            // into bf16 hi + lo by the packed conversions (two entries per instruction, already in operand order)
            float e[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = mine[j] ? __builtin_bit_cast(float, v4[j]) - dd[j] : 0.f;
            pr.af.x = blk_cvt_pk(e[0], e[1]);
            pr.af.y = blk_cvt_pk(e[2], e[3]);
            l[0] = e[0] - __builtin_bit_cast(float, pr.af.x << 16);
            l[1] = e[1] - __builtin_bit_cast(float, pr.af.x & 0xFFFF0000u);
            l[2] = e[2] - __builtin_bit_cast(float, pr.af.y << 16);
            l[3] = e[3] - __builtin_bit_cast(float, pr.af.y & 0xFFFF0000u);
            pr.af.z = blk_cvt_pk(l[0], l[1]);
            pr.af.w = blk_cvt_pk(l[2], l[3]);
            if (a.evals != nullptr && exists) {        // the owner of an entry stores its e as hi | lo << 16
                uint32_t* ev = a.evals + (int64_t)(gfirst + k) * 16 + 4 * g;
                if (mine[0]) ev[0] = __builtin_amdgcn_perm(pr.af.z, pr.af.x, 0x05040100u);
                if (mine[1]) ev[1] = __builtin_amdgcn_perm(pr.af.z, pr.af.x, 0x07060302u);
                if (mine[2]) ev[2] = __builtin_amdgcn_perm(pr.af.w, pr.af.y, 0x05040100u);
                if (mine[3]) ev[3] = __builtin_amdgcn_perm(pr.af.w, pr.af.y, 0x07060302u);
            }
        }
        // B: lane r of the group addresses the 4 columns 4 (r & 3) .. + 3 (of the chunk) of entry 4g + (r >> 2); the
        // transposed read hands lane i the values of column i of the group's four entries
        const uint32_t ilb = (il4 >> (8 * (sub >> 2))) & 0xFFu;
        pr.rowb = buf + ilb * BLK_PITCH + (sub & 3) * 8;
        return pr;
    };
    // the eight transposed reads of a group (one per 16 columns) and its eight matrix-core steps; BEHIND = LDS reads issued
    // after the group's own (the next group's: they stay in flight).  The four values a lane receives serve the hi AND the lo
    // K slots of the operand: two register moves per step (an LDS read instead measured slower: the LDS pipe is the busier one)
    struct Reads {
        s16x4 t0, t1, t2, t3, t4, t5, t6, t7;
    };
    auto reads = [&](const unsigned char* rowb) {
        Reads r;
        r.t0 = lds_read_tr16_b64<0>(rowb);   r.t1 = lds_read_tr16_b64<32>(rowb);  r.t2 = lds_read_tr16_b64<64>(rowb);
        r.t3 = lds_read_tr16_b64<96>(rowb);  r.t4 = lds_read_tr16_b64<128>(rowb); r.t5 = lds_read_tr16_b64<160>(rowb);
        r.t6 = lds_read_tr16_b64<192>(rowb); r.t7 = lds_read_tr16_b64<224>(rowb);
        return r;
    };
    auto steps = [&](auto behind, const u32x4& af, Reads& r) {
        constexpr int B = decltype(behind)::value;
        const bf16x8 afb = __builtin_bit_cast(bf16x8, af);
        auto one = [&](auto cc, s16x4& t) {
            constexpr int C = decltype(cc)::value;
            if (VAR == 1) {
                if (C == 7) { lds_wait<B>(t); acc[0][0] += (float)(r.t0[0] + r.t1[0] + r.t2[0] + r.t3[0] + r.t4[0] + r.t5[0] + r.t6[0] + t[0]) + __builtin_bit_cast(float, af.x); }
                return;
            }
            lds_wait<B + 7 - C>(t);
            acc[C] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afb, __builtin_bit_cast(bf16x8, (s16x8){t[0], t[1], t[2], t[3], t[0], t[1], t[2], t[3]}), acc[C], 0, 0, 0);
        };
        one(std::integral_constant<int, 0>(), r.t0);
        one(std::integral_constant<int, 1>(), r.t1);
        one(std::integral_constant<int, 2>(), r.t2);
        one(std::integral_constant<int, 3>(), r.t3);
        one(std::integral_constant<int, 4>(), r.t4);
        one(std::integral_constant<int, 5>(), r.t5);
        one(std::integral_constant<int, 6>(), r.t6);
        one(std::integral_constant<int, 7>(), r.t7);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 8> I8;
    // prep with a compile-time K (the DPP lane as an immediate, no switch)
    auto prep_k = [&](auto kk, const Meta& mt, int gfirst, const unsigned char* buf, bool exists) {
        constexpr int K = decltype(kk)::value;
        return prep(K, mt, gfirst, buf, exists);
    };
    auto batch = [&](const Meta& mt, int gfirst, int ng, const unsigned char* buf) {       // ng: scalar, 1 .. 16
        if (VAR == 0 || VAR == 6) {
            // unrolled; the vector-ALU preparation of group K + 1 is issued between the reads and the steps of group K (it is
            // computed whether or not that group exists: the metadata loads clamp to the cell's last group)
            Prep pc = prep_k(I0(), mt, gfirst, buf, true);
            static_for<16>([&](auto kk) {
                constexpr int K = decltype(kk)::value;
                if (K < ng) {                                                  // (scalar)
                    Reads rc = reads(pc.rowb);
                    const u32x4 af = pc.af;
                    if (K + 1 < 16) pc = prep_k(std::integral_constant<int, (K + 1 < 16 ? K + 1 : 15)>(), mt, gfirst, buf, K + 1 < ng);
                    steps(I0(), af, rc);
                }
            });
        } else if (MODE == BLK_APPLY && VAR == 5) {
            // software-pipelined by two: the preparation and the reads of group k + 1 go out before the steps of group k
            Prep pa = prep(0, mt, gfirst, buf), pb;
            Reads ra = reads(pa.rowb), rb;
            for (int k = 0; k < ng; k += 2) {
                if (k + 1 < ng) {
                    pb = prep(k + 1, mt, gfirst, buf);
                    rb = reads(pb.rowb);
                    steps(I8(), pa.af, ra);
                } else {
                    steps(I0(), pa.af, ra);
                    break;
                }
                if (k + 2 < ng) {
                    pa = prep(k + 2, mt, gfirst, buf);
                    ra = reads(pa.rowb);
                    steps(I8(), pb.af, rb);
                } else {
                    steps(I0(), pb.af, rb);
                    break;
                }
            }
        } else {
            for (int k = 0; k < ng; ++k) {
                Prep pc;
                if (VAR == 2) { pc.af = mt.v; pc.rowb = buf + ((mt.il + k) & 0xFFu) * BLK_PITCH + (sub & 3) * 8; }
                else pc = prep(k, mt, gfirst, buf);
                if (VAR == 3) {
                    const s16x4 t = {(short)lane, (short)k, 1, 2};
                    const bf16x8 afb = __builtin_bit_cast(bf16x8, pc.af);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afb, __builtin_bit_cast(bf16x8, (s16x8){t[0], t[1], t[2], t[3], t[0], t[1], t[2], t[3]}), acc[c], 0, 0, 0);
                } else {
                    Reads rc = reads(pc.rowb);
                    steps(I0(), pc.af, rc);
                }
            }
        }
    };

    const int nb = b1 - b0;
    int cpv0 = 0, cpv1 = 0;                     // cellptr[b0 + 64 w + lane], [.. + 1] of the current window of 64 blocks
    auto load_window = [&](int i0) {
        const int i = i0 + lane;
        cpv0 = (live && i <= nb) ? cp[b0 + i] : 0;
        cpv1 = (live && i + 1 <= nb) ? cp[b0 + i + 1] : 0;
    };

    Meta cur, nxt;
    if (nb > 0) {
        load_window(0);
        const int g0 = __builtin_amdgcn_readlane(cpv0, 0), g1 = __builtin_amdgcn_readlane(cpv1, 0);
        cur = load_meta(g0, g1 - g0 < 16 ? g1 - g0 : 16);
        dma_block(b0, 0);
    } else {
        cur = load_meta(0, 0);
    }
    nxt = cur;
    __syncthreads();

    for (int i = 0; i < nb; ++i) {
        const unsigned char* buf = lds + (i & 1) * BLK_BUF;
        const int g0 = __builtin_amdgcn_readlane(cpv0, i & 63), g1 = __builtin_amdgcn_readlane(cpv1, i & 63);
        // the next cell's metadata and the next block's rows are requested before this cell is worked on
        if (i + 1 < nb) {
            if (((i + 1) & 63) == 0) load_window(i + 1);
            const int n0 = __builtin_amdgcn_readlane(cpv0, (i + 1) & 63), n1 = __builtin_amdgcn_readlane(cpv1, (i + 1) & 63);
            nxt = load_meta(n0, n1 - n0 < 16 ? n1 - n0 : 16);
            dma_block(b0 + i + 1, (i + 1) & 1);
        }
        // the first 16 groups of the cell from the metadata requested one block ago (complete since the last barrier: the
        // compiler needs no vmcnt wait here, the LDS-DMA and the next cell's metadata stay in flight under the work) ...
        if (g0 < g1) batch(cur, g0, g1 - g0 < 16 ? g1 - g0 : 16, buf);
        // ... a cell of more than 16 groups (denser than 1 entry in 16): the rest batch by batch, each behind its own load
        for (int gf = g0 + 16; gf < g1; gf += 16) {
            const int ng = g1 - gf < 16 ? g1 - gf : 16;
            const Meta more = load_meta(gf, ng);
            batch(more, gf, ng, buf);
        }
        cur = nxt;
        __syncthreads();                                                   // (drains this wave's LDS-DMA: vmcnt(0))
    }

    // out[part][strip * 16 + 4 g + r][16 c + sub] = acc[c][r]
    if (live) {
        float* dst = a.out + (int64_t)part * a.part_stride;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t o = strip * BLK_OS + 4 * g + r;
            if (o < a.n_out) {
#pragma unroll
                for (int c = 0; c < 8; ++c) dst[o * a.ld_out + 16 * c + sub] = acc[c][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same pass with TWO strips per wave (8 waves of up to 256 registers per workgroup instead of 16 of 128): a wave works
// through the cells of its two strips group by group in lockstep -- preparation and reads of strip A's group, preparation and
// reads of strip B's group, steps of A, steps of B -- so that every group's LDS latency and vector-ALU preparation lie under
// the matrix-core steps of the other strip's group: two independent chains per wave, which the 16-wave form has no
// registers for (its pipelined variants spilled).  Same lists, same arithmetic and summation order per strip, same bits.
// ------------------------------------------------------------------------------------------
constexpr int BLK2_WAVES = 8;

template <int MODE>
__global__ __launch_bounds__(512) void blk2_pass_kernel(BlkArgs a) {
    HIP_DYNAMIC_SHARED(unsigned char, lds)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7;
    const int lg = a.parts >= 8 ? 3 : a.parts >= 4 ? 2 : a.parts >= 2 ? 1 : 0;
    const int part = xcd & (a.parts - 1);
    const int64_t ob = (int64_t)(blockIdx.x >> 3) * (8 >> lg) + (xcd >> lg);
    if (ob * BLK_WGR >= a.n_out) return;                                   // (workgroup-uniform)
    const int b0 = part * a.blk_per_part;
    const int b1 = b0 + a.blk_per_part < a.nblk ? b0 + a.blk_per_part : a.nblk;
    const int nb = b1 - b0;
    const int64_t strip0 = ob * (BLK_WGR / BLK_OS) + 2 * wv;
    bool live[2];
    const int* cp[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        live[s] = (strip0 + s) * BLK_OS < a.n_out;
        cp[s] = a.cellptr + (live[s] ? strip0 + s : 0) * (int64_t)(a.nblk + 1);
    }
    const unsigned char* fi = (const unsigned char*)a.Fi;
    const int64_t ldb = a.ldi * 2;
    auto dma_block = [&](int blk, int buf) {
        const int64_t base = (int64_t)blk * BLK_IB;
        const unsigned char* src = fi + base * ldb;
        const int64_t left = a.n_in - base;
        const int rmax = (int)(left < BLK_IB ? left : BLK_IB) - 1;
#pragma unroll
        for (int i = 0; i < BLK_DMA / BLK2_WAVES; ++i) {                   // 72 pieces, 9 per wave
            const int q = wv + BLK2_WAVES * i;
            const int slot = q * 64 + lane;
            int row = (slot * 3641) >> 16;
            int w = slot - row * 18;
            w = w < 16 ? w : 15;
            row = row < rmax ? row : rmax;
            const uint32_t off = (uint32_t)row * (uint32_t)ldb + (uint32_t)w * 16u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                             (__attribute__((address_space(3))) void*)(lds + buf * BLK_BUF + q * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[2][8];
    u32x4 fo[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) fo[s][k] = u32x4{0u, 0u, 0u, 0u};
        if (MODE == BLK_RESIDUAL && live[s]) {
            int64_t o = (strip0 + s) * BLK_OS + sub;
            o = o < a.n_out ? o : a.n_out - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) fo[s][k] = *(const u32x4*)(a.Fo + o * a.ldo + 32 * k + 8 * g);
        }
    }
    struct Meta {
        uint32_t il, ol;
        u32x4 v;
    };
    auto load_meta = [&](int first, int count) {
        Meta m;
        m.il = 0u;
        m.ol = 0xFFFFFFFFu;
        m.v = u32x4{0u, 0u, 0u, 0u};
        if (count > 0) {
            const int64_t gi = first + (sub < count ? sub : count - 1);
            const uint32_t* mp = a.meta + gi * 8;
            m.il = mp[g];
            m.ol = mp[4 + g];
            m.v = MODE == BLK_APPLY ? *(const u32x4*)(a.evals + gi * 16 + 4 * g) : *(const u32x4*)(a.rvals + gi * 16 + 4 * g);
        }
        return m;
    };
    struct Prep {
        u32x4 af;
        const unsigned char* rowb;
    };
    struct Reads {
        s16x4 t0, t1, t2, t3, t4, t5, t6, t7;
    };
    auto prep = [&](auto kk, auto ss, const Meta& mt, int gfirst, const unsigned char* buf) {
        constexpr int K = decltype(kk)::value, S = decltype(ss)::value;
        Prep pr;
        const uint32_t ol4 = row_bcast_u<K>(mt.ol), il4 = row_bcast_u<K>(mt.il);
        const uint32_t v4[4] = {row_bcast_u<K>(mt.v.x), row_bcast_u<K>(mt.v.y), row_bcast_u<K>(mt.v.z), row_bcast_u<K>(mt.v.w)};
        bool mine[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[j] = ((ol4 >> (8 * j)) & 0xFFu) == (uint32_t)sub;
        if (MODE == BLK_APPLY) {
            uint32_t mk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mk[j] = mine[j] ? v4[j] : 0u;
            pr.af.x = __builtin_amdgcn_perm(mk[1], mk[0], 0x05040100u);
            pr.af.y = __builtin_amdgcn_perm(mk[3], mk[2], 0x05040100u);
            pr.af.z = __builtin_amdgcn_perm(mk[1], mk[0], 0x07060302u);
            pr.af.w = __builtin_amdgcn_perm(mk[3], mk[2], 0x07060302u);
        } else {
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)il4, 0), s1 = (uint32_t)__builtin_amdgcn_readlane((int)il4, 16),
                           s2 = (uint32_t)__builtin_amdgcn_readlane((int)il4, 32), s3 = (uint32_t)__builtin_amdgcn_readlane((int)il4, 48);
            const int wa = sub >> 2;
            const uint32_t ila4 = wa == 0 ? s0 : wa == 1 ? s1 : wa == 2 ? s2 : s3;
            const uint32_t ila = (ila4 >> (8 * (sub & 3))) & 0xFFu;
            const unsigned char* rowa = buf + ila * BLK_PITCH + g * 16;
            f32x4 dd = {0.f, 0.f, 0.f, 0.f};
            u32x4 a0 = lds_read_b128<0>(rowa), a1 = lds_read_b128<64>(rowa), a2 = lds_read_b128<128>(rowa), a3 = lds_read_b128<192>(rowa);
            lds_wait<3>(a0);
            dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, fo[S][0]), dd, 0, 0, 0);
            lds_wait<2>(a1);
            dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, fo[S][1]), dd, 0, 0, 0);
            lds_wait<1>(a2);
            dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, fo[S][2]), dd, 0, 0, 0);
            lds_wait<0>(a3);
            dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3), __builtin_bit_cast(bf16x8, fo[S][3]), dd, 0, 0, 0);
            float e[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = mine[j] ? __builtin_bit_cast(float, v4[j]) - dd[j] : 0.f;
            pr.af.x = blk_cvt_pk(e[0], e[1]);
            pr.af.y = blk_cvt_pk(e[2], e[3]);
            l[0] = e[0] - __builtin_bit_cast(float, pr.af.x << 16);
            l[1] = e[1] - __builtin_bit_cast(float, pr.af.x & 0xFFFF0000u);
            l[2] = e[2] - __builtin_bit_cast(float, pr.af.y << 16);
            l[3] = e[3] - __builtin_bit_cast(float, pr.af.y & 0xFFFF0000u);
            pr.af.z = blk_cvt_pk(l[0], l[1]);
            pr.af.w = blk_cvt_pk(l[2], l[3]);
            if (a.evals != nullptr) {
                uint32_t* ev = a.evals + (int64_t)(gfirst + K) * 16 + 4 * g;
                if (mine[0]) ev[0] = __builtin_amdgcn_perm(pr.af.z, pr.af.x, 0x05040100u);
                if (mine[1]) ev[1] = __builtin_amdgcn_perm(pr.af.z, pr.af.x, 0x07060302u);
                if (mine[2]) ev[2] = __builtin_amdgcn_perm(pr.af.w, pr.af.y, 0x05040100u);
                if (mine[3]) ev[3] = __builtin_amdgcn_perm(pr.af.w, pr.af.y, 0x07060302u);
            }
        }
        const uint32_t ilb = (il4 >> (8 * (sub >> 2))) & 0xFFu;
        pr.rowb = buf + ilb * BLK_PITCH + (sub & 3) * 8;
        return pr;
    };
    auto reads = [&](const unsigned char* rowb) {
        Reads r;
        r.t0 = lds_read_tr16_b64<0>(rowb);   r.t1 = lds_read_tr16_b64<32>(rowb);  r.t2 = lds_read_tr16_b64<64>(rowb);
        r.t3 = lds_read_tr16_b64<96>(rowb);  r.t4 = lds_read_tr16_b64<128>(rowb); r.t5 = lds_read_tr16_b64<160>(rowb);
        r.t6 = lds_read_tr16_b64<192>(rowb); r.t7 = lds_read_tr16_b64<224>(rowb);
        return r;
    };
    auto steps = [&](auto ss, auto behind, const u32x4& af, Reads& r) {
        constexpr int S = decltype(ss)::value, B = decltype(behind)::value;
        const bf16x8 afb = __builtin_bit_cast(bf16x8, af);
        auto one = [&](auto cc, s16x4& t) {
            constexpr int C = decltype(cc)::value;
            lds_wait<B + 7 - C>(t);
            acc[S][C] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afb, __builtin_bit_cast(bf16x8, (s16x8){t[0], t[1], t[2], t[3], t[0], t[1], t[2], t[3]}), acc[S][C], 0, 0, 0);
        };
        one(std::integral_constant<int, 0>(), r.t0);
        one(std::integral_constant<int, 1>(), r.t1);
        one(std::integral_constant<int, 2>(), r.t2);
        one(std::integral_constant<int, 3>(), r.t3);
        one(std::integral_constant<int, 4>(), r.t4);
        one(std::integral_constant<int, 5>(), r.t5);
        one(std::integral_constant<int, 6>(), r.t6);
        one(std::integral_constant<int, 7>(), r.t7);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 8> I8;

    int cpv0[2] = {0, 0}, cpv1[2] = {0, 0};
    auto load_window = [&](int i0) {
        const int i = i0 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            cpv0[s] = (live[s] && i <= nb) ? cp[s][b0 + i] : 0;
            cpv1[s] = (live[s] && i + 1 <= nb) ? cp[s][b0 + i + 1] : 0;
        }
    };
    Meta cur[2], nxt[2];
    if (nb > 0) {
        load_window(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int g0 = __builtin_amdgcn_readlane(cpv0[s], 0), g1 = __builtin_amdgcn_readlane(cpv1[s], 0);
            cur[s] = load_meta(g0, g1 - g0 < 16 ? g1 - g0 : 16);
        }
        dma_block(b0, 0);
    } else {
        cur[0] = cur[1] = load_meta(0, 0);
    }
    nxt[0] = cur[0];
    nxt[1] = cur[1];
    __syncthreads();

    for (int i = 0; i < nb; ++i) {
        const unsigned char* buf = lds + (i & 1) * BLK_BUF;
        int g0[2], g1[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            g0[s] = __builtin_amdgcn_readlane(cpv0[s], i & 63);
            g1[s] = __builtin_amdgcn_readlane(cpv1[s], i & 63);
        }
        if (i + 1 < nb) {
            if (((i + 1) & 63) == 0) load_window(i + 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int n0 = __builtin_amdgcn_readlane(cpv0[s], (i + 1) & 63), n1 = __builtin_amdgcn_readlane(cpv1[s], (i + 1) & 63);
                nxt[s] = load_meta(n0, n1 - n0 < 16 ? n1 - n0 : 16);
            }
            dma_block(b0 + i + 1, (i + 1) & 1);
        }
        const int nga = g1[0] - g0[0] < 16 ? g1[0] - g0[0] : 16, ngb = g1[1] - g0[1] < 16 ? g1[1] - g0[1] : 16;
        // the first 16 groups of the two cells in lockstep
        static_for<16>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            const bool ha = K < nga, hb = K < ngb;                         // (scalar)
            if (ha && hb) {
                const Prep pa = prep(kk, I0(), cur[0], g0[0], buf);
                Reads ra = reads(pa.rowb);
                const Prep pb = prep(kk, I1(), cur[1], g0[1], buf);
                Reads rb = reads(pb.rowb);
                steps(I0(), I8(), pa.af, ra);
                steps(I1(), I0(), pb.af, rb);
            } else if (ha) {
                const Prep pa = prep(kk, I0(), cur[0], g0[0], buf);
                Reads ra = reads(pa.rowb);
                steps(I0(), I0(), pa.af, ra);
            } else if (hb) {
                const Prep pb = prep(kk, I1(), cur[1], g0[1], buf);
                Reads rb = reads(pb.rowb);
                steps(I1(), I0(), pb.af, rb);
            }
        });
        // cells of more than 16 groups: the rest strip by strip, batch by batch, each behind its own load
        static_for<2>([&](auto ss) {
            constexpr int S = decltype(ss)::value;
            for (int gf = g0[S] + 16; gf < g1[S]; gf += 16) {
                const int ng = g1[S] - gf < 16 ? g1[S] - gf : 16;
                const Meta more = load_meta(gf, ng);
                static_for<16>([&](auto kk) {
                    constexpr int K = decltype(kk)::value;
                    if (K < ng) {
                        const Prep pc = prep(kk, ss, more, gf, buf);
                        Reads rc = reads(pc.rowb);
                        steps(ss, I0(), pc.af, rc);
                    }
                });
            }
        });
        cur[0] = nxt[0];
        cur[1] = nxt[1];
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (live[s]) {
            float* dst = a.out + (int64_t)part * a.part_stride;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t o = (strip0 + s) * BLK_OS + 4 * g + r;
                if (o < a.n_out) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) dst[o * a.ld_out + 16 * c + sub] = acc[s][c][r];
                }
            }
        }
    }
}


}  // namespace skf
