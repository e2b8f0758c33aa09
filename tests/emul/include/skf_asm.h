// Host stand-ins for scikit-fusion_amd/csrc/skf_asm.h (the SIMT emulator build force-includes this file first, so the
// product header's include guard skips the inline-assembly forms).  Same names and signatures; the emulator models the
// instruction behind each builtin's name and has no outstanding-operation counters, so the waits are empty.
#ifndef SKF_ASM_H_
#define SKF_ASM_H_
#include <hip/hip_runtime.h>
#include <cstdint>

namespace skf {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16_b64(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + OFF));
}
__device__ __forceinline__ void lds_tr_wait(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
}

template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(const unsigned char* p) {
    return *(const u32x4*)(p + OFF);
}
template <int CNT, typename T>
__device__ __forceinline__ void lds_wait(T& x) {
}
template <int CNT, typename T, typename U>
__device__ __forceinline__ void lds_wait(T& x, U& y) {
}
template <typename T>
__device__ __forceinline__ void lds_claim(T& x) {
}
__device__ __forceinline__ uint32_t global_load_u32(const void* p) {
    return *(const uint32_t*)p;
}
template <int CNT>
__device__ __forceinline__ void vm_wait(uint32_t& x) {
}
__device__ __forceinline__ void lds_write_b128(u32x4* p, u32x4 v) {
    *p = v;
}

}  // namespace skf

#endif  // SKF_ASM_H_
