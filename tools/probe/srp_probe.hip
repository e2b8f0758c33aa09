// srp_probe.hip -- stand-alone throughput probe of the known-entry passes of skf_known.h (sparse-residual DFMC):
// random lists of `per` entries per outer object, gathered rows of width w, 1 / 2 / 4 / 8 parts pinned to XCDs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I scikit-fusion_amd/csrc tools/probe/srp_probe.hip -o tools/probe/srp_probe
//   tools/probe/srp_probe <n_out> <n_in> <per> <w> <bf16|f32>     (SRP_TUNED=1 | SRP_V6=1: the tuned kernels; SRP_PARTS=n)
#include "skf_known.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

using namespace skf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <typename TG>
static void run(int64_t n_out, int64_t n_in, int per, int w) {
    const int64_t nnz = n_out * per;
    std::vector<int> idx((size_t)nnz);
    const double stride = (double)n_in / per;
    for (int64_t o = 0; o < n_out; ++o)
        for (int k = 0; k < per; ++k) {
            int64_t c = (int64_t)(k * stride + (rng() % 1000000) * 1e-6 * stride);
            idx[(size_t)(o * per + k)] = (int)(c < n_in ? c : n_in - 1);
        }
    std::vector<TG> fo((size_t)n_out * w), fi((size_t)(n_in + 1) * w);          // (one all-zero row behind the gathered matrix)
    for (auto& v : fo) v = sizeof(TG) == 2 ? (TG)f32_to_bf16_rne((rng() % 1000) * 1e-3f) : (TG)((rng() % 1000) * 1e-3f);
    for (auto& v : fi) v = sizeof(TG) == 2 ? (TG)f32_to_bf16_rne((rng() % 1000) * 1e-3f) : (TG)((rng() % 1000) * 1e-3f);
    for (int t = 0; t < w; ++t) fi[(size_t)n_in * w + t] = (TG)0;
    std::vector<float> rv((size_t)nnz, 0.5f);
    int* d_idx; float *d_rv, *d_ev, *d_out; TG *d_fo, *d_fi; int64_t* d_ptr;
    CK(hipMalloc(&d_idx, nnz * 4)); CK(hipMalloc(&d_rv, nnz * 4)); CK(hipMalloc(&d_ev, nnz * 4));
    CK(hipMalloc(&d_fo, fo.size() * sizeof(TG))); CK(hipMalloc(&d_fi, fi.size() * sizeof(TG)));
    CK(hipMalloc(&d_out, (size_t)8 * n_out * w * 4)); CK(hipMalloc(&d_ptr, (size_t)(n_out * 8 + 1) * 8));
    CK(hipMemcpy(d_idx, idx.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rv, rv.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_fo, fo.data(), fo.size() * sizeof(TG), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_fi, fi.data(), fi.size() * sizeof(TG), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int only_parts = getenv("SRP_PARTS") ? atoi(getenv("SRP_PARTS")) : 0;     // SRP_PARTS=n: that many parts only (PMC passes)
    for (int parts = 1; parts <= 8; parts *= 2) {
        if (only_parts && parts != only_parts) continue;
        // segment pointers: the entries of an outer object are ascending, a part = a contiguous range of the inner index
        std::vector<int64_t> ptr((size_t)(n_out * parts + 1));
        const int64_t pw = ((n_in + parts - 1) / parts + 63) / 64 * 64;
        for (int64_t o = 0; o < n_out; ++o) {
            int64_t q = o * per;
            for (int p = 0; p < parts; ++p) {
                ptr[(size_t)(o * parts + p)] = q;
                while (q < (o + 1) * per && idx[(size_t)q] < (p + 1) * pw) ++q;
            }
        }
        ptr[(size_t)(n_out * parts)] = nnz;
        CK(hipMemcpy(d_ptr, ptr.data(), ptr.size() * 8, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 2; ++mode) {
            for (int store = 0; store < (mode == 0 ? 2 : 1); ++store) {
                SrpArgs<TG, float> a;
                memset(&a, 0, sizeof a);
                a.ptr = d_ptr; a.idx = d_idx; a.rvals = d_rv; a.evals = (mode == 1 || store) ? d_ev : nullptr;
                a.Fo = d_fo; a.Fi = d_fi; a.out = d_out; a.ldo = w; a.ldi = w; a.ld_out = w;
                a.part_stride = n_out * w; a.n_out = n_out; a.w = w; a.parts = parts; a.mode = mode;
                a.zero_off = (uint32_t)(n_in * w * sizeof(TG));
                const int per_round = 8 / parts;
                const int64_t blocks_per_part = (n_out + 3) / 4;
                const int grid = (int)((blocks_per_part + per_round - 1) / per_round * 8);
                auto launch = [&]() {
                    constexpr int VE = 16 / (int)sizeof(TG);
                    const int gl = w / VE;
                    if constexpr (sizeof(TG) == 2) {
                        if (getenv("SRP_V6") && (gl == 16 || gl == 32)) {          // srp_bf16_v6_kernel (w = 128 / 256)
#define PROBE_V6(CPL_) do { if (mode == 0) hipLaunchKernelGGL((srp_bf16_v6_kernel<CPL_, SRP_RESIDUAL>), dim3(grid), dim3(256), 0, 0, a); \
                           else hipLaunchKernelGGL((srp_bf16_v6_kernel<CPL_, SRP_APPLY>), dim3(grid), dim3(256), 0, 0, a); } while (0)
                            if (gl == 16) PROBE_V6(1); else PROBE_V6(2);
                            return;
                        }
                        if (getenv("SRP_TUNED")) {
#define PROBE_SRP(GL_) do { if (mode == 0) hipLaunchKernelGGL((srp_bf16_kernel<GL_, SRP_RESIDUAL>), dim3(grid), dim3(256), 0, 0, a); \
                           else hipLaunchKernelGGL((srp_bf16_kernel<GL_, SRP_APPLY>), dim3(grid), dim3(256), 0, 0, a); } while (0)
                            if (gl == 8) PROBE_SRP(8);
                            else if (gl == 16) PROBE_SRP(16);
                            else if (gl == 32) PROBE_SRP(32);
                            return;
                        }
                    }
                    if (gl == 8) hipLaunchKernelGGL((srp_vec_kernel<TG, float, 8>), dim3(grid), dim3(256), 0, 0, a);
                    else if (gl == 16) hipLaunchKernelGGL((srp_vec_kernel<TG, float, 16>), dim3(grid), dim3(256), 0, 0, a);
                    else if (gl == 32) hipLaunchKernelGGL((srp_vec_kernel<TG, float, 32>), dim3(grid), dim3(256), 0, 0, a);
                    else if (gl == 64) hipLaunchKernelGGL((srp_vec_kernel<TG, float, 64>), dim3(grid), dim3(256), 0, 0, a);
                    else hipLaunchKernelGGL((srp_any_kernel<TG, float>), dim3(grid), dim3(256), 0, 0, a);
                };
                launch();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int it = 0; it < 5; ++it) launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= 5;
                {   // checksum of the pass (parts summed) and of the stored residuals: the flavours must agree
                    std::vector<float> ho((size_t)parts * n_out * w), he((size_t)nnz);
                    CK(hipMemcpy(ho.data(), d_out, ho.size() * 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(he.data(), d_ev, he.size() * 4, hipMemcpyDeviceToHost));
                    double so = 0, se = 0;
                    for (float v : ho) so += (double)v * (double)v;
                    for (float v : he) se += (double)v * (double)v;
                    printf("  checksum out %.9e residuals %.9e\n", so, se);
                }
                printf("n_out %lld n_in %lld per %d w %d %s parts %d mode %s%s: %.3f ms  gathered %.2f GB -> %.2f TB/s\n",
                       (long long)n_out, (long long)n_in, per, w, sizeof(TG) == 2 ? "bf16" : "f32", parts,
                       mode == 0 ? "residual" : "apply", store ? "+store" : "", ms, nnz * (double)w * sizeof(TG) / 1e9,
                       nnz * (double)w * sizeof(TG) / 1e9 / ms);
            }
        }
    }
    CK(hipFree(d_idx)); CK(hipFree(d_rv)); CK(hipFree(d_ev)); CK(hipFree(d_fo)); CK(hipFree(d_fi)); CK(hipFree(d_out)); CK(hipFree(d_ptr));
}

int main(int argc, char** argv) {
    if (argc < 6) { printf("usage: srp_probe n_out n_in per w bf16|f32\n"); return 1; }
    const int64_t n_out = atoll(argv[1]), n_in = atoll(argv[2]);
    const int per = atoi(argv[3]), w = atoi(argv[4]);
    if (!strcmp(argv[5], "bf16")) run<uint16_t>(n_out, n_in, per, w);
    else run<float>(n_out, n_in, per, w);
    return 0;
}
