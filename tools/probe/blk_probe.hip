// blk_probe.hip -- stand-alone probe of the BLOCKED known-entry passes (skf_blocked.h) against the per-entry gather
// kernels of skf_known.h (srp_bf16_v6_kernel) on the same random lists: times, agreement of the outputs and of the stored
// residuals, and a host fp64 check on sampled outer objects.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I scikit-fusion_amd/csrc -I tools/probe -I include tools/probe/blk_probe.hip -o tools/probe/blk_probe
//   tools/probe/blk_probe <n_out> <n_in> <per> [arrange 0|1|2] [parts]      (w = 128, bf16)
// arrange: 0 = entries of a cell in list order; 1 = positions matched to the rows' bank residues, what does not fit goes
// to the free positions of the cell's ceil(n / 16) groups; 2 = residues strictly (extra groups instead of conflicts)
#include "skf_blocked.h"   // (tools/probe: not part of the product build)

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

using namespace skf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

struct Ent { int ol, il; float r; };

// positions of a cell's entries: returns the number of groups, fills pos[e] (group * 16 + position)
static int arrange_cell(const std::vector<Ent>& ents, int mode, std::vector<int>& pos) {
    const int n = (int)ents.size();
    pos.assign(n, -1);
    if (n == 0) return 0;
    int groups = (n + 15) / 16;
    if (mode == 0) {
        for (int e = 0; e < n; ++e) pos[e] = e;
        return groups;
    }
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mode == 2) {
        int c2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (const Ent& e : ents) ++c2[e.il & 7];
        for (int r = 0; r < 8; ++r) groups = std::max(groups, (c2[r] + 1) / 2);
    }
    std::vector<char> used((size_t)groups * 16, 0);
    std::vector<int> over;
    for (int e = 0; e < n; ++e) {
        const int res = ents[e].il & 7, t = cnt[res]++;
        const int grp = t >> 1, p = res + 8 * (t & 1);
        if (grp < groups) { pos[e] = grp * 16 + p; used[pos[e]] = 1; }
        else over.push_back(e);
    }
    int f = 0;
    for (int e : over) {
        while (used[f]) ++f;
        pos[e] = f;
        used[f] = 1;
    }
    return groups;
}

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: blk_probe n_out n_in per [arrange] [parts]\n"); return 1; }
    const int64_t n_out = atoll(argv[1]), n_in = atoll(argv[2]);
    const int per = atoi(argv[3]);
    const int arrange = argc > 4 ? atoi(argv[4]) : 1;
    const int parts = argc > 5 ? atoi(argv[5]) : 8;
    const int w = BLK_W;
    const int64_t nnz = n_out * per;
    std::vector<int> idx((size_t)nnz);
    const double stride = (double)n_in / per;
    for (int64_t o = 0; o < n_out; ++o)
        for (int k = 0; k < per; ++k) {
            int64_t c = (int64_t)(k * stride + (rng() % 1000000) * 1e-6 * stride);
            idx[(size_t)(o * per + k)] = (int)(c < n_in ? c : n_in - 1);
        }
    std::vector<uint16_t> fo((size_t)n_out * w), fi((size_t)(n_in + 1) * w);
    for (auto& v : fo) v = f32_to_bf16_rne((rng() % 1000) * 1e-3f - 0.3f);
    for (auto& v : fi) v = f32_to_bf16_rne((rng() % 1000) * 1e-3f - 0.4f);
    for (int t = 0; t < w; ++t) fi[(size_t)n_in * w + t] = 0;
    std::vector<float> rv((size_t)nnz);
    for (auto& v : rv) v = (rng() % 1000) * 1e-2f;

    // ---- blocked lists (host) ----
    const int64_t strips = (n_out + BLK_OS - 1) / BLK_OS;
    const int nblk = (int)((n_in + BLK_IB - 1) / BLK_IB);
    std::vector<int> cellptr((size_t)strips * (nblk + 1));
    std::vector<uint32_t> meta;
    std::vector<float> brv;
    std::vector<int64_t> bsrc;                     // list position of every blocked slot (-1 = padding)
    {
        std::vector<int64_t> cur(BLK_OS);
        std::vector<Ent> ents;
        std::vector<int64_t> src;
        std::vector<int> pos;
        int64_t groups_total = 0, pad = 0;
        for (int64_t s = 0; s < strips; ++s) {
            for (int r = 0; r < BLK_OS; ++r) cur[r] = (s * BLK_OS + r) * per;
            for (int b = 0; b < nblk; ++b) {
                cellptr[(size_t)s * (nblk + 1) + b] = (int)groups_total;
                ents.clear(); src.clear();
                for (int r = 0; r < BLK_OS; ++r) {
                    const int64_t o = s * BLK_OS + r;
                    if (o >= n_out) continue;
                    const int64_t end = (o + 1) * per;
                    while (cur[r] < end && idx[(size_t)cur[r]] < (int64_t)(b + 1) * BLK_IB) {
                        ents.push_back(Ent{r, (int)(idx[(size_t)cur[r]] - (int64_t)b * BLK_IB), rv[(size_t)cur[r]]});
                        src.push_back(cur[r]);
                        ++cur[r];
                    }
                }
                const int gc = arrange_cell(ents, arrange, pos);
                const size_t base = meta.size();
                meta.resize(base + (size_t)gc * 8, 0u);
                brv.resize(brv.size() + (size_t)gc * 16, 0.f);
                bsrc.resize(bsrc.size() + (size_t)gc * 16, -1);
                for (int g = 0; g < gc; ++g)
                    for (int j = 4; j < 8; ++j) meta[base + (size_t)g * 8 + j] = 0xFFFFFFFFu;          // outer index: padding
                for (size_t e = 0; e < ents.size(); ++e) {
                    const int g = pos[e] >> 4, p = pos[e] & 15;
                    uint32_t& wi = meta[base + (size_t)g * 8 + (p >> 2)];
                    uint32_t& wo = meta[base + (size_t)g * 8 + 4 + (p >> 2)];
                    wi = (wi & ~(0xFFu << (8 * (p & 3)))) | ((uint32_t)ents[e].il << (8 * (p & 3)));
                    wo = (wo & ~(0xFFu << (8 * (p & 3)))) | ((uint32_t)ents[e].ol << (8 * (p & 3)));
                    brv[(size_t)(groups_total + g) * 16 + p] = ents[e].r;
                    bsrc[(size_t)(groups_total + g) * 16 + p] = src[e];
                }
                pad += (int64_t)gc * 16 - (int64_t)ents.size();
                groups_total += gc;
            }
            cellptr[(size_t)s * (nblk + 1) + nblk] = (int)groups_total;
        }
        printf("blocked lists: %lld strips x %d blocks, %lld groups, %.1f %% padding, arrange %d\n", (long long)strips, nblk,
               (long long)groups_total, 100.0 * pad / (16.0 * groups_total), arrange);
    }
    const int64_t groups = (int64_t)meta.size() / 8;

    // ---- device buffers ----
    int* d_idx; float *d_rv, *d_ev, *d_out, *d_out2; uint16_t *d_fo, *d_fi; int64_t* d_ptr;
    int* d_cell; uint32_t *d_meta, *d_bev; float* d_brv;
    CK(hipMalloc(&d_idx, nnz * 4)); CK(hipMalloc(&d_rv, nnz * 4)); CK(hipMalloc(&d_ev, nnz * 4));
    CK(hipMalloc(&d_fo, fo.size() * 2)); CK(hipMalloc(&d_fi, fi.size() * 2));
    CK(hipMalloc(&d_out, (size_t)8 * n_out * w * 4)); CK(hipMalloc(&d_out2, (size_t)8 * n_out * w * 4));
    CK(hipMalloc(&d_ptr, (size_t)(n_out * 8 + 1) * 8));
    CK(hipMalloc(&d_cell, cellptr.size() * 4)); CK(hipMalloc(&d_meta, meta.size() * 4)); CK(hipMalloc(&d_bev, (size_t)groups * 64));
    CK(hipMalloc(&d_brv, (size_t)groups * 64));
    CK(hipMemcpy(d_idx, idx.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rv, rv.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_fo, fo.data(), fo.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_fi, fi.data(), fi.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cell, cellptr.data(), cellptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_brv, brv.data(), brv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_bev, 0, (size_t)groups * 64));
    CK(hipFuncSetAttribute((const void*)blk_pass_kernel<BLK_APPLY>, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS));
    CK(hipFuncSetAttribute((const void*)blk_pass_kernel<BLK_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // ---- reference passes: srp_bf16_v6_kernel, 8 parts ----
    const int rparts = 8;
    {
        std::vector<int64_t> ptr((size_t)(n_out * rparts + 1));
        const int64_t pw = ((n_in + rparts - 1) / rparts + 63) / 64 * 64;
        for (int64_t o = 0; o < n_out; ++o) {
            int64_t q = o * per;
            for (int p = 0; p < rparts; ++p) {
                ptr[(size_t)(o * rparts + p)] = q;
                while (q < (o + 1) * per && idx[(size_t)q] < (p + 1) * pw) ++q;
            }
        }
        ptr[(size_t)(n_out * rparts)] = nnz;
        CK(hipMemcpy(d_ptr, ptr.data(), ptr.size() * 8, hipMemcpyHostToDevice));
    }
    auto time_it = [&](const char* what, auto&& launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < 5; ++it) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %.3f ms   (%.1f ps per entry)\n", what, ms / 5, ms / 5 * 1e9 / (double)nnz);
        return ms / 5;
    };
    SrpArgs<uint16_t, float> sa;
    memset(&sa, 0, sizeof sa);
    sa.ptr = d_ptr; sa.idx = d_idx; sa.rvals = d_rv; sa.evals = d_ev; sa.Fo = d_fo; sa.Fi = d_fi; sa.out = d_out;
    sa.ldo = w; sa.ldi = w; sa.ld_out = w; sa.part_stride = n_out * w; sa.n_out = n_out; sa.w = w; sa.parts = rparts;
    sa.zero_off = (uint32_t)(n_in * w * 2);
    const int sgrid = (int)(((n_out + 3) / 4 + (8 / rparts) - 1) / (8 / rparts) * 8);
    sa.mode = SRP_RESIDUAL;
    time_it("v6 residual + store (8 parts)", [&] { hipLaunchKernelGGL((srp_bf16_v6_kernel<1, SRP_RESIDUAL>), dim3(sgrid), dim3(256), 0, 0, sa); });

    BlkArgs ba;
    memset(&ba, 0, sizeof ba);
    ba.cellptr = d_cell; ba.meta = d_meta; ba.rvals = d_brv; ba.evals = d_bev; ba.Fo = d_fo; ba.Fi = d_fi; ba.out = d_out2;
    ba.ldo = w; ba.ldi = w; ba.ld_out = w; ba.part_stride = n_out * w; ba.n_out = n_out; ba.n_in = n_in;
    ba.nblk = nblk; ba.parts = parts; ba.blk_per_part = (nblk + parts - 1) / parts;
    const int64_t obs = (n_out + BLK_WGR - 1) / BLK_WGR;
    const int bgrid = (int)((obs + (8 / parts) - 1) / (8 / parts) * 8);
    time_it("blocked residual + store", [&] { hipLaunchKernelGGL((blk_pass_kernel<BLK_RESIDUAL>), dim3(bgrid), dim3(1024), BLK_LDS, 0, ba); });

    auto compare = [&](const char* what, int pa, int pb) {
        std::vector<float> A((size_t)pa * n_out * w), B((size_t)pb * n_out * w);
        CK(hipMemcpy(A.data(), d_out, A.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(B.data(), d_out2, B.size() * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, worst = 0;
        for (int64_t e = 0; e < n_out * w; ++e) {
            double x = 0, y = 0;
            for (int p = 0; p < pa; ++p) x += A[(size_t)p * n_out * w + e];
            for (int p = 0; p < pb; ++p) y += B[(size_t)p * n_out * w + e];
            num += (x - y) * (x - y); den += x * x;
            worst = std::max(worst, fabs(x - y));
        }
        printf("%s: blocked vs v6 outputs, relative (Frobenius) %.3e, max abs diff %.3e (rms %.3e)\n", what, sqrt(num / den), worst,
               sqrt(den / (n_out * w)));
        return B;
    };
    std::vector<float> outB = compare("residual pass", rparts, parts);
    {   // stored residuals: blocked (hi + lo) against v6 (f32), and a host fp64 check of sampled outer objects
        std::vector<float> he((size_t)nnz);
        std::vector<uint32_t> hb((size_t)groups * 16);
        CK(hipMemcpy(he.data(), d_ev, he.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), d_bev, hb.size() * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        int64_t seen = 0;
        for (size_t s = 0; s < hb.size(); ++s) {
            if (bsrc[s] < 0) continue;
            const double x = he[(size_t)bsrc[s]], y = blk_unpack_hi_lo(hb[s]);
            num += (x - y) * (x - y); den += x * x; ++seen;
        }
        printf("stored residuals: %lld entries, relative difference %.3e\n", (long long)seen, sqrt(num / den));
        if (getenv("BLK_DEBUG")) {
            int shown = 0;
            for (size_t s = 0; s < hb.size() && shown < 40; ++s) {
                if (bsrc[s] < 0) continue;
                const double x = he[(size_t)bsrc[s]], y = blk_unpack_hi_lo(hb[s]);
                const size_t grp = s / 16; const int p_ = (int)(s % 16);
                const uint32_t il = (meta[grp * 8 + (p_ >> 2)] >> (8 * (p_ & 3))) & 0xFF, ol = (meta[grp * 8 + 4 + (p_ >> 2)] >> (8 * (p_ & 3))) & 0xFF;
                printf("  group %zu pos %2d il %3u ol %2u  v6 %10.4f  blocked %10.4f %s\n", grp, p_, il, ol, x, y, fabs(x - y) > 1e-2 * fabs(x) + 1e-3 ? "<<" : "");
                ++shown;
            }
        }
        double worst = 0;
        for (int t = 0; t < 32; ++t) {
            const int64_t o = (int64_t)(rng() % (uint64_t)n_out);
            std::vector<double> ref(w, 0.0);
            for (int k = 0; k < per; ++k) {
                const int64_t q = o * per + k, i = idx[(size_t)q];
                double x = 0;
                for (int c = 0; c < w; ++c) x += (double)bf16_to_f32(fo[(size_t)o * w + c]) * (double)bf16_to_f32(fi[(size_t)i * w + c]);
                const double e = rv[(size_t)q] - x;
                for (int c = 0; c < w; ++c) ref[c] += e * (double)bf16_to_f32(fi[(size_t)i * w + c]);
            }
            double num2 = 0, den2 = 0;
            for (int c = 0; c < w; ++c) {
                double y = 0;
                for (int p = 0; p < parts; ++p) y += outB[(size_t)p * n_out * w + (size_t)o * w + c];
                num2 += (y - ref[c]) * (y - ref[c]); den2 += ref[c] * ref[c];
            }
            worst = std::max(worst, sqrt(num2 / den2));
        }
        printf("blocked residual pass vs host fp64 on 32 outer objects: worst relative %.3e\n", worst);
    }
    sa.mode = SRP_APPLY;
    time_it("v6 apply (8 parts)", [&] { hipLaunchKernelGGL((srp_bf16_v6_kernel<1, SRP_APPLY>), dim3(sgrid), dim3(256), 0, 0, sa); });
    ba.rvals = nullptr;
    time_it("blocked apply", [&] { hipLaunchKernelGGL((blk_pass_kernel<BLK_APPLY>), dim3(bgrid), dim3(1024), BLK_LDS, 0, ba); });
    compare("apply pass", rparts, parts);
    {   // two strips per wave (8 waves of 256 registers)
        CK(hipFuncSetAttribute((const void*)blk2_pass_kernel<BLK_APPLY>, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS));
        CK(hipFuncSetAttribute((const void*)blk2_pass_kernel<BLK_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS));
        CK(hipMemset(d_out2, 0, (size_t)8 * n_out * w * 4));
        time_it("blocked apply, two strips per wave", [&] { hipLaunchKernelGGL((blk2_pass_kernel<BLK_APPLY>), dim3(bgrid), dim3(512), BLK_LDS, 0, ba); });
        compare("apply pass, two strips per wave", rparts, parts);
        ba.rvals = d_brv;
        CK(hipMemset(d_out2, 0, (size_t)8 * n_out * w * 4));
        time_it("blocked residual + store, two strips per wave", [&] { hipLaunchKernelGGL((blk2_pass_kernel<BLK_RESIDUAL>), dim3(bgrid), dim3(512), BLK_LDS, 0, ba); });
        sa.mode = SRP_RESIDUAL;
        hipLaunchKernelGGL((srp_bf16_v6_kernel<1, SRP_RESIDUAL>), dim3(sgrid), dim3(256), 0, 0, sa);
        CK(hipDeviceSynchronize());
        compare("residual pass, two strips per wave", rparts, parts);
        ba.rvals = nullptr;
    }
    if (getenv("BLK_BOUNDS")) {        // bound-finding variants of the apply pass (wrong results by construction)
        CK(hipFuncSetAttribute((const void*)blk_pass_kernel<BLK_APPLY, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS));
        CK(hipFuncSetAttribute((const void*)blk_pass_kernel<BLK_APPLY, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS));
        CK(hipFuncSetAttribute((const void*)blk_pass_kernel<BLK_APPLY, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS));
        CK(hipFuncSetAttribute((const void*)blk_pass_kernel<BLK_APPLY, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS));
        time_it("blocked apply, pipelined by two (spills)", [&] { hipLaunchKernelGGL((blk_pass_kernel<BLK_APPLY, 5>), dim3(bgrid), dim3(1024), BLK_LDS, 0, ba); });
        time_it("blocked apply, no matrix-core steps", [&] { hipLaunchKernelGGL((blk_pass_kernel<BLK_APPLY, 1>), dim3(bgrid), dim3(1024), BLK_LDS, 0, ba); });
        time_it("blocked apply, no vector-ALU preparation", [&] { hipLaunchKernelGGL((blk_pass_kernel<BLK_APPLY, 2>), dim3(bgrid), dim3(1024), BLK_LDS, 0, ba); });
        time_it("blocked apply, no LDS reads", [&] { hipLaunchKernelGGL((blk_pass_kernel<BLK_APPLY, 3>), dim3(bgrid), dim3(1024), BLK_LDS, 0, ba); });
    }
    return 0;
}
