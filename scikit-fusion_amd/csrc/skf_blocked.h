// skf_blocked.h -- the known-entry passes of DFMC as BLOCKED SDDMM + SpMM on the matrix cores (round 6).
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

template <int MODE>
__global__ __launch_bounds__(1024) void blk_pass_kernel(BlkArgs a) {
    HIP_DYNAMIC_SHARED(unsigned char, lds)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
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
#pragma unroll
        for (int i = 0; i < (BLK_DMA + BLK_WAVES - 1) / BLK_WAVES; ++i) {
            const int q = wv + BLK_WAVES * i;
            if (q < BLK_DMA) {                                             // (wave-uniform)
                const int slot = q * 64 + lane;
                const int row = (slot * 3641) >> 16;                       // slot / 18 for slot < 4608
                int w = slot - row * 18;
                w = w < 16 ? w : 15;
                int64_t grow = base + row;
                grow = grow < a.n_in ? grow : a.n_in - 1;                  // rows past the end: never referenced by an entry
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(fi + grow * ldb + w * 16),
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
    // outer indices of its four entries, their values, and the inner indices (BLK_RESIDUAL: of all sixteen) -- and a group's
    // turn hands them to the row of 16 lanes by DPP row broadcasts (no LDS, no reload per group)
    struct Meta {
        u32x4 il;            // BLK_APPLY: .x = the four inner indices of lane group g; BLK_RESIDUAL: all sixteen
        uint32_t ol;
        u32x4 v;
    };
    auto load_meta = [&](int first, int count) {
        Meta m;
        m.il = u32x4{0u, 0u, 0u, 0u};
        m.ol = 0xFFFFFFFFu;
        m.v = u32x4{0u, 0u, 0u, 0u};
        if (count > 0) {
            const int64_t gi = first + (sub < count ? sub : count - 1);
            const uint32_t* mp = a.meta + gi * 8;
            if (MODE == BLK_APPLY) m.il.x = mp[g];
            else m.il = *(const u32x4*)mp;
            m.ol = mp[4 + g];
            m.v = MODE == BLK_APPLY ? *(const u32x4*)(a.evals + gi * 16 + 4 * g) : *(const u32x4*)(a.rvals + gi * 16 + 4 * g);
        }
        return m;
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
        cur = load_meta(g0, live ? (g1 - g0 < 16 ? g1 - g0 : 16) : 0);
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
        int n0 = 0, n1 = 0;
        if (i + 1 < nb) {
            if (((i + 1) & 63) == 0) load_window(i + 1);
            n0 = __builtin_amdgcn_readlane(cpv0, (i + 1) & 63);
            n1 = __builtin_amdgcn_readlane(cpv1, (i + 1) & 63);
            nxt = load_meta(n0, live ? (n1 - n0 < 16 ? n1 - n0 : 16) : 0);
            dma_block(b0 + i + 1, (i + 1) & 1);
        }
        int gfirst = g0;
        const int gend = live ? g1 : g0;
        while (gfirst < gend) {
            const int ng = gend - gfirst < 16 ? gend - gfirst : 16;
            auto group = [&](auto kk) {
                constexpr int K = decltype(kk)::value;
                const uint32_t ol4 = row_bcast_u<K>(cur.ol);
                const u32x4 v4 = {row_bcast_u<K>(cur.v.x), row_bcast_u<K>(cur.v.y), row_bcast_u<K>(cur.v.z), row_bcast_u<K>(cur.v.w)};
                uint32_t il4, mk[4];
                bool mine[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) mine[j] = ((ol4 >> (8 * j)) & 0xFFu) == (uint32_t)sub;
                if (MODE == BLK_APPLY) {
                    il4 = row_bcast_u<K>(cur.il.x);
#pragma unroll
                    for (int j = 0; j < 4; ++j) mk[j] = mine[j] ? v4[j] : 0u;
                } else {
                    const u32x4 il16 = {row_bcast_u<K>(cur.il.x), row_bcast_u<K>(cur.il.y), row_bcast_u<K>(cur.il.z), row_bcast_u<K>(cur.il.w)};
                    il4 = g == 0 ? il16.x : g == 1 ? il16.y : g == 2 ? il16.z : il16.w;
                    const int wa = sub >> 2;
                    const uint32_t ila4 = wa == 0 ? il16.x : wa == 1 ? il16.y : wa == 2 ? il16.z : il16.w;
                    const uint32_t ila = (ila4 >> (8 * (sub & 3))) & 0xFFu;
                    // SDDMM: A = the 16 gathered rows (lane (m = sub, g): columns 32 k + 8 g .. + 7 of entry sub's row)
                    const unsigned char* rowa = buf + ila * BLK_PITCH + g * 16;
                    u32x4 a0 = lds_read_b128<0>(rowa), a1 = lds_read_b128<64>(rowa), a2 = lds_read_b128<128>(rowa),
                          a3 = lds_read_b128<192>(rowa);
                    f32x4 dd = {0.f, 0.f, 0.f, 0.f};
                    lds_wait<3>(a0);
                    dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, fo[0]), dd, 0, 0, 0);
                    lds_wait<2>(a1);
                    dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, fo[1]), dd, 0, 0, 0);
                    lds_wait<1>(a2);
                    dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, fo[2]), dd, 0, 0, 0);
                    lds_wait<0>(a3);
                    dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3), __builtin_bit_cast(bf16x8, fo[3]), dd, 0, 0, 0);
                    // lane (o = sub, g), element j: <Fi[entry 4g + j], Fo[o]>; the entry's own outer object keeps r - x
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float e = __builtin_bit_cast(float, v4[j]) - dd[j];
                        mk[j] = mine[j] ? blk_pack_hi_lo(e) : 0u;
                    }
                    if (a.evals != nullptr) {
                        uint32_t* ev = a.evals + (int64_t)(gfirst + K) * 16 + 4 * g;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (mine[j]) ev[j] = mk[j];
                    }
                }
                // SpMM: A[outer = sub][K slot (g, j)] = hi (j < 4) / lo (j >= 4) of entry 4g + (j & 3) where it is sub's own
                u32x4 af;
                af.x = (mk[0] & 0xFFFFu) | (mk[1] << 16);
                af.y = (mk[2] & 0xFFFFu) | (mk[3] << 16);
                af.z = (mk[0] >> 16) | (mk[1] & 0xFFFF0000u);
                af.w = (mk[2] >> 16) | (mk[3] & 0xFFFF0000u);
                // B: lane r of the group addresses the 4 columns 4 (r & 3) .. + 3 (of the chunk) of entry 4g + (r >> 2); the
                // transposed read hands lane i the values of column i of the group's four entries
                const uint32_t ilb = (il4 >> (8 * (sub >> 2))) & 0xFFu;
                const unsigned char* rowb = buf + ilb * BLK_PITCH + (sub & 3) * 8;
                s16x4 t0 = lds_read_tr16_b64<0>(rowb), t1 = lds_read_tr16_b64<32>(rowb), t2 = lds_read_tr16_b64<64>(rowb),
                      t3 = lds_read_tr16_b64<96>(rowb), t4 = lds_read_tr16_b64<128>(rowb), t5 = lds_read_tr16_b64<160>(rowb),
                      t6 = lds_read_tr16_b64<192>(rowb), t7 = lds_read_tr16_b64<224>(rowb);
                const bf16x8 afb = __builtin_bit_cast(bf16x8, af);
                auto step = [&](auto cc, s16x4& t) {
                    constexpr int C = decltype(cc)::value;
                    lds_wait<7 - C>(t);
                    const s16x8 b = {t[0], t[1], t[2], t[3], t[0], t[1], t[2], t[3]};
                    acc[C] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afb, __builtin_bit_cast(bf16x8, b), acc[C], 0, 0, 0);
                };
                step(std::integral_constant<int, 0>(), t0);
                step(std::integral_constant<int, 1>(), t1);
                step(std::integral_constant<int, 2>(), t2);
                step(std::integral_constant<int, 3>(), t3);
                step(std::integral_constant<int, 4>(), t4);
                step(std::integral_constant<int, 5>(), t5);
                step(std::integral_constant<int, 6>(), t6);
                step(std::integral_constant<int, 7>(), t7);
            };
            group(std::integral_constant<int, 0>());
            if (ng > 1) group(std::integral_constant<int, 1>());
            if (ng > 2) group(std::integral_constant<int, 2>());
            if (ng > 3) group(std::integral_constant<int, 3>());
            if (ng > 4) group(std::integral_constant<int, 4>());
            if (ng > 5) group(std::integral_constant<int, 5>());
            if (ng > 6) group(std::integral_constant<int, 6>());
            if (ng > 7) group(std::integral_constant<int, 7>());
            if (ng > 8) group(std::integral_constant<int, 8>());
            if (ng > 9) group(std::integral_constant<int, 9>());
            if (ng > 10) group(std::integral_constant<int, 10>());
            if (ng > 11) group(std::integral_constant<int, 11>());
            if (ng > 12) group(std::integral_constant<int, 12>());
            if (ng > 13) group(std::integral_constant<int, 13>());
            if (ng > 14) group(std::integral_constant<int, 14>());
            if (ng > 15) group(std::integral_constant<int, 15>());
            gfirst += ng;
            if (gfirst < gend) cur = load_meta(gfirst, gend - gfirst < 16 ? gend - gfirst : 16);     // (a cell of more than 16 groups)
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

}  // namespace skf
