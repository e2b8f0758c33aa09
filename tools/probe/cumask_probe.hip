// cumask_probe: which physical CUs does a stream created with hipExtStreamCreateWithCUMask(mask) run on?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/cumask_probe.hip -o tools/probe/cumask_probe
//   tools/probe/cumask_probe "248-255" "31,63,95,127,159,191,223,255" ""
// Every argument is a list of mask BITS TO CLEAR (as SKF_MAIN_CU_DROP takes it); 8192 workgroups of one wave record
// (XCC_ID, SE_ID, CU_ID) from the hardware registers; printed: distinct CUs per XCC.  Answers how mask bits map onto the
// 8 XCDs x 32 CUs of an MI355X before the engine's contraction stream is masked (round 5).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <vector>

__global__ void where_am_i(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup alive for a while so that the launch spreads over every CU the mask allows
    uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, ncu);
    const int nwg = 8192;
    uint32_t* d;
    hipMalloc(&d, nwg * 2 * sizeof(uint32_t));
    std::vector<uint32_t> h(nwg * 2);
    for (int a = 1; a < argc || a == 1; ++a) {
        const char* drop = a < argc ? argv[a] : "";
        std::vector<uint32_t> mask((ncu + 31) / 32, 0xffffffffu);
        if (ncu % 32) mask.back() = (1u << (ncu % 32)) - 1u;
        const char* c = drop;
        int dropped = 0;
        while (*c) {
            char* end = nullptr;
            long lo = strtol(c, &end, 10), hi = lo;
            if (end == c) break;
            c = end;
            if (*c == '-') { hi = strtol(c + 1, &end, 10); c = end; }
            for (long k = lo; k <= hi && k < ncu; ++k)
                if (k >= 0 && ((mask[k / 32] >> (k % 32)) & 1u)) { mask[k / 32] &= ~(1u << (k % 32)); ++dropped; }
            if (*c == ',') ++c;
        }
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            printf("drop [%s]: hipExtStreamCreateWithCUMask failed\n", drop);
            continue;
        }
        hipMemsetAsync(d, 0xff, nwg * 2 * sizeof(uint32_t), st);
        hipLaunchKernelGGL(where_am_i, dim3(nwg), dim3(64), 0, st, d, 20000);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, nwg * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
        std::set<uint32_t> per_xcc[16];
        for (int b = 0; b < nwg; ++b) {
            const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        int total = 0;
        printf("drop [%s] (%d bits cleared): CUs seen per XCC:", drop, dropped);
        for (int x = 0; x < 16; ++x)
            if (!per_xcc[x].empty()) { printf(" x%d=%zu", x, per_xcc[x].size()); total += (int)per_xcc[x].size(); }
        printf("  total %d\n", total);
        hipStreamDestroy(st);
    }
    hipFree(d);
    return 0;
}
