// skf_asm.h -- the gfx950 instructions the kernels issue as inline assembly, with the waits that claim their results.
//
// The compiler's waitcnt pass treats the builtin forms of these LDS reads as possible readers of every LDS-DMA in
// flight and serialises the ring of gemm_bf16_v2_kernel with s_waitcnt vmcnt(0); it knows nothing about the assembly
// forms, so the K loop counts its own outstanding operations: lds_wait<N>() = s_waitcnt lgkmcnt(N), vm_wait<N>() =
// s_waitcnt vmcnt(N), each tied to the registers it releases.
// (tests/emul/include/skf_asm.h holds host stand-ins with the same names for the SIMT emulator build, which force-includes
// it ahead of this file; this file is the only form the product build sees.)
#ifndef SKF_ASM_H_
#define SKF_ASM_H_
#include <hip/hip_runtime.h>
#include <cstdint>

namespace skf {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ds_read_b64_tr_b16 with an immediate byte offset; the results are claimed with lds_tr_wait() -- lgkmcnt(0) tied to
// the result registers -- before the first use
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16_b64(const unsigned char* p) {
    s16x4 r;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ void lds_tr_wait(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}

// ds_read_b128 in the same inline-assembly form (the fragment pipeline of gemm_bf16_v2_kernel counts its own
// outstanding LDS reads: lds_wait<N>() = s_waitcnt lgkmcnt(N) tied to the registers it releases)
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(const unsigned char* p) {
    u32x4 r;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int CNT, typename T>
__device__ __forceinline__ void lds_wait(T& x) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(CNT));
}
template <int CNT, typename T, typename U>
__device__ __forceinline__ void lds_wait(T& x, U& y) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(CNT));
}
// the register holds a value that an earlier lds_wait has released: ties its consumers behind that wait
template <typename T>
__device__ __forceinline__ void lds_claim(T& x) {
    asm volatile("" : "+v"(x));
}
// the bitmap loads and expansion stores of the ABITS flavour, in the same form (vm_wait<N>() = s_waitcnt vmcnt(N))
__device__ __forceinline__ uint32_t global_load_u32(const void* p) {
    uint32_t r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int CNT>
__device__ __forceinline__ void vm_wait(uint32_t& x) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(CNT));
}
__device__ __forceinline__ void lds_write_b128(u32x4* p, u32x4 v) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)p;
    asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

}  // namespace skf

#endif  // SKF_ASM_H_
