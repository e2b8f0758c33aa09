// Probe of ds_read_b64_tr_b16 on gfx950: prints, for every lane, which LDS halfwords the
// instruction returns when lane l passes the byte address 8*l (lds[i] = i).  The kernels'
// transposed fragment reads (and the host emulator's model of the instruction) rely on
//     lane i of a 16-lane group, element j  <-  the value addressed by lane 4*j + (i>>2), its element i&3
// Also prints the XCC id of the first workgroups (blockIdx -> XCD placement, for information).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int* xcc) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4));
    if (blockIdx.x == 0)
        for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
    if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) ;  // HW_REG_XCC_ID bits 0..3
}
int main() {
    short* d; int* x;
    hipMalloc(&d, 256 * 2); hipMalloc(&x, 64 * 4);
    hipLaunchKernelGGL(k, dim3(32), dim3(64), 0, 0, d, x);
    short h[256]; int hx[32];
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    hipMemcpy(hx, x, 128, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int g = l >> 4, i = l & 15;
            const int expect = (16 * g + 4 * j + (i >> 2)) * 4 + (i & 3);
            printf(" %4d%s", h[l * 4 + j], h[l * 4 + j] == expect ? "" : "!");
            bad += h[l * 4 + j] != expect;
        }
        printf("\n");
    }
    printf("tr16_b64 model %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    printf("xcc ids of blocks 0..31:");
    for (int b = 0; b < 32; ++b) printf(" %d", hx[b]);
    printf("\n");
    return 0;
}
