// Sustained MFMA rate of the whole chip with real (random) and all-zero operands, 16x16x32 vs 32x32x16 bf16:
// how much of the gap between the contraction and the 2.5 PFLOP/s peak is the power-managed clock.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_probe.hip -o tools/probe/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ char pad[1024];
    const int tid = blockIdx.x * 512 + threadIdx.x;
    u32x4 fa[4], fb[8];
    for (int i = 0; i < 4; ++i) fa[i] = src[(tid * 12 + i) & 0xFFFFF];
    for (int j = 0; j < 8; ++j) fb[j] = src[(tid * 12 + 4 + j) & 0xFFFFF];
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[4][8];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[2][4];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // same flops per iteration: two k16 steps, fragments (a: 2 per step, b: 4 per step)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[2 * ks + i]), __builtin_bit_cast(bf16x8, fb[4 * ks + j]), acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (s == 12345.678f) out[tid] = s + pad[threadIdx.x & 1023];
}

int main() {
    const int n = 1 << 20;
    std::vector<u32x4> h(n);
    u32x4* d; float* o;
    hipMalloc(&d, n * sizeof(u32x4)); hipMalloc(&o, 256 * 4 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int k = 0; k < n; ++k)
            for (int q = 0; q < 4; ++q) {
                // bf16 pairs of U[0,1): random mantissa, exponent 0x3F00..0x3F7F
                unsigned lo = 0x3F00u | (rand() & 0xFF), hi = 0x3F00u | (rand() & 0xFF);
                h[k][q] = mode == 0 ? (lo | (hi << 16)) : 0u;
            }
        hipMemcpy(d, h.data(), n * sizeof(u32x4), hipMemcpyHostToDevice);
        for (int shape = 16; shape <= 32; shape += 16) {
            const int iters = 4000, blocks = 256 * 4;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(512), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(512), 0, 0, d, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)blocks * 8 /*waves*/ * iters * 32 /*mfma*/ * 16384.0;
                if (rep == 2) printf("%s operands, %dx%d: %.3f ms  %.0f TFLOP/s\n", mode == 0 ? "random U[0,1)" : "all-zero", shape, shape, ms, flops / ms * 1e-9);
            }
        }
    }
    return 0;
}
