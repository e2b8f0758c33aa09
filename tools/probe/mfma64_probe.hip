// Sustained rate of the f64 / f32 matrix-core instructions of the whole chip with nothing else in the loop (operands in
// registers, independent accumulators): the ceiling the f64-accumulated products of an iteration (Gram = G^T G, W = G_i^T P,
// sweep inverse) and the f32 side updates are priced against.  Random and all-zero operands, 1 .. 4 waves per SIMD, and f64
// FMAs on the vector ALU beside the matrix cores (do the two pipes add up?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma64_probe.hip -o tools/probe/mfma64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: v_mfma_f64_16x16x4_f64, 8 accumulators;  MODE 1: v_mfma_f32_32x32x2_f32, 4 accumulators
// MODE 2: f64 MFMA with f64 vector FMAs interleaved (8 MFMA : 32 v_fma_f64);  MODE 3: the vector FMAs alone
template <int MODE>
__global__ __launch_bounds__(256) void loop(const double* __restrict__ src, double* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    double a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = src[(tid * 6 + i) & 0xFFFFF];
    for (int j = 0; j < 4; ++j) b[j] = src[(tid * 6 + 2 + j) & 0xFFFFF];
    double s = 0.0;
    if constexpr (MODE == 0 || MODE == 2 || MODE == 3) {
        f64x4 acc[2][4];
        double v[32];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < 32; ++q) v[q] = a[q & 1] * 1e-3 * q;
        for (int it = 0; it < iters; ++it) {
            if (MODE != 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (MODE >= 2) {
#pragma unroll
                for (int q = 0; q < 32; ++q) v[q] = __builtin_fma(v[q], b[q & 3], a[q & 1]);
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
        for (int q = 0; q < 32; ++q) s += v[q];
    } else {
        f32x16 acc[4];
        const float af[2] = {(float)a[0], (float)a[1]}, bf[2] = {(float)b[0], (float)b[1]};
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[2 * i + j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    if (s == 12345.678) out[tid] = s;
}

template <int MODE>
static void run(const double* d, double* o, const char* what, const char* ops, double flops_per_wave_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; ++wps) {            // waves per SIMD: blocks of 4 waves, `wps` blocks per CU
        const int iters = 2000, blocks = 256 * wps;
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-34s %-14s %d waves/SIMD: %8.3f ms  %6.1f TFLOP/s\n", what, ops, wps, ms, (double)blocks * 4 * iters * flops_per_wave_iter / ms * 1e-9);
    }
}

int main() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    double *d, *o;
    hipMalloc(&d, n * 8); hipMalloc(&o, 256 * 4 * 256 * 8);
    for (int mode = 0; mode < 2; ++mode) {
        for (int k = 0; k < n; ++k) h[k] = mode == 0 ? (double)rand() / RAND_MAX : 0.0;
        hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
        const char* ops = mode == 0 ? "random U[0,1)" : "all-zero";
        run<0>(d, o, "v_mfma_f64_16x16x4_f64", ops, 8 * 2048.0);
        run<1>(d, o, "v_mfma_f32_32x32x2_f32", ops, 8 * 4096.0);
        run<3>(d, o, "v_fma_f64 (vector ALU)", ops, 32 * 128.0);
        run<2>(d, o, "f64 MFMA + v_fma_f64 interleaved", ops, 8 * 2048.0 + 32 * 128.0);
    }
    return 0;
}
