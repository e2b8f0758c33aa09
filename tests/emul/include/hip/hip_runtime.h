// =====================================================================================
// HOST SIMT EMULATOR  --  TEST INFRASTRUCTURE ONLY (never part of the product build)
//
// A stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel sources under
// scikit-fusion_amd/csrc/ be compiled with host clang++ (-x c++) and executed on the CPU:
// every GPU thread of a workgroup is a ucontext fiber; __syncthreads() and the wave-level
// collectives (__shfl*, the MFMA builtins) are rendezvous points between fibers.  The
// point is to check index arithmetic, tile/boundary handling, LDS staging and the barrier
// structure of the kernels in the build container, which has no GPU.  Timing, caches,
// async copies and memory-model effects are NOT modelled.
//
// The MFMA builtins are emulated with the gfx950 lane<->element maps given in
// /opt/skills/guides/cdna_hip_programming.md (section 3):
//   mfma_f32_32x32x2f32 : A[i=l&31][k=l>>5]  B[k=l>>5][j=l&31]
//                         D reg r: row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31
//   mfma_f32_16x16x4f32 : A[i=l&15][k=l>>4]  B[k=l>>4][j=l&15]   D reg r: row=4*(l>>4)+r, col=l&15
//   mfma_f64_16x16x4f64 : A[i=l&15][k=l>>4]  B[k=l>>4][j=l&15]   D reg r: row=(l>>4)+4*r, col=l&15
//   mfma_f32_32x32x16_bf16 : A[i=l&31][k=8*(l>>5)+e]  B[k=8*(l>>5)+e][j=l&31]  D as 32x32x2f32
//   mfma_f32_16x16x32_bf16 : A[i=l&15][k=8*(l>>4)+e]  B[k=8*(l>>4)+e][j=l&15]
//                         D reg r: row=4*(l>>4)+r, col=l&15
// Set SKF_EMUL_REVERSE=1 to schedule fibers in reverse order (flushes out missing barriers
// that a forward order hides).
// =====================================================================================
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct simt_uint3 { unsigned x, y, z; };

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h,
                                   hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
// no graph support in the emulator: capture reports failure and the library stays eager
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = nullptr; return hipErrorInvalidValue; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
#define hipEventDisableTiming 0x2
#define hipStreamNonBlocking 0x1

namespace simt {

enum State { READY = 0, AT_BARRIER = 1, WAVE_WAIT = 2, DONE = 3 };
constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 192 * 1024;
constexpr size_t XCHG_BYTES = 160;          // per-lane exchange slot for wave collectives

// Fiber switch.  swapcontext() saves and restores the signal mask with two system calls per switch -- a third of the CPU
// suite's time; on x86-64 the switch is the callee-saved registers and the stack pointer, nothing else is live across a
// call (the kernels do not touch MXCSR / the x87 control word).  Other hosts keep ucontext.
#if defined(__x86_64__)
#define SIMT_FAST_SWITCH 1
__attribute__((naked, noinline)) static void fiber_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
    asm volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
#else
#define SIMT_FAST_SWITCH 0
#endif

struct Fiber {
    ucontext_t ctx;
    void* sp = nullptr;
    simt_uint3 tid;
    int linear = 0;
    int state = READY;
    char* stack = nullptr;
};

struct WaveSync {
    int arrived = 0;
    unsigned gen = 0;
    alignas(16) unsigned char xchg[WAVE][XCHG_BYTES];
};

inline Fiber* g_cur = nullptr;
inline simt_uint3 g_blockIdx{0, 0, 0};
inline dim3 g_blockDim, g_gridDim;
inline ucontext_t g_sched;
inline void* g_sched_sp = nullptr;
inline std::vector<Fiber> g_fibers;
inline std::vector<WaveSync> g_waves;
inline std::function<void()>* g_body = nullptr;
inline int g_nthreads = 0;
inline int g_progress = 0;
inline unsigned char* g_dyn_smem = nullptr;
inline size_t g_dyn_cap = 0;
inline void set_dynamic_smem(size_t bytes) {
    if (bytes > g_dyn_cap) {
        free(g_dyn_smem);
        g_dyn_smem = (unsigned char*)aligned_alloc(256, (bytes + 255) / 256 * 256);
        g_dyn_cap = bytes;
    }
}

inline void die(const char* msg) {
    fprintf(stderr, "[simt-emul] fatal: %s (block %u,%u,%u)\n", msg, g_blockIdx.x, g_blockIdx.y, g_blockIdx.z);
    abort();
}

inline void yield_to_scheduler() {
#if SIMT_FAST_SWITCH
    fiber_switch(&g_cur->sp, g_sched_sp);
#else
    swapcontext(&g_cur->ctx, &g_sched);
#endif
}

inline void block_barrier() {
    g_cur->state = AT_BARRIER;
    yield_to_scheduler();
}

inline int wave_lanes(int wave) {           // lanes that exist in this wave
    int lo = wave * WAVE, hi = lo + WAVE;
    if (hi > g_nthreads) hi = g_nthreads;
    return hi - lo;
}

// rendezvous of all lanes of the calling lane's wave
inline void wave_sync() {
    Fiber* me = g_cur;
    int w = me->linear / WAVE;
    WaveSync& ws = g_waves[w];
    unsigned gen = ws.gen;
    if (++ws.arrived == wave_lanes(w)) {
        ws.arrived = 0;
        ws.gen++;
        for (int l = 0; l < wave_lanes(w); ++l) {
            Fiber& f = g_fibers[w * WAVE + l];
            if (f.state == WAVE_WAIT) f.state = READY;
        }
        return;
    }
    while (ws.gen == gen) {
        me->state = WAVE_WAIT;
        yield_to_scheduler();
    }
}

inline void trampoline() {
    (*g_body)();
    g_cur->state = DONE;
    yield_to_scheduler();
    die("resumed a finished fiber");
}

inline void run_block(std::function<void()>& body) {
    static const bool reverse = getenv("SKF_EMUL_REVERSE") && atoi(getenv("SKF_EMUL_REVERSE")) != 0;
    const int n = g_nthreads;
    if ((int)g_fibers.size() < n) {
        size_t old = g_fibers.size();
        g_fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) g_fibers[i].stack = (char*)malloc(STACK_BYTES);
    }
    int nw = (n + WAVE - 1) / WAVE;
    if ((int)g_waves.size() < nw) g_waves.resize(nw);
    for (int w = 0; w < nw; ++w) { g_waves[w].arrived = 0; g_waves[w].gen = 0; }
    g_body = &body;
    for (int t = 0; t < n; ++t) {
        Fiber& f = g_fibers[t];
        f.linear = t;
        f.tid.x = t % g_blockDim.x;
        f.tid.y = (t / g_blockDim.x) % g_blockDim.y;
        f.tid.z = t / (g_blockDim.x * g_blockDim.y);
        f.state = READY;
#if SIMT_FAST_SWITCH
        // a fresh stack as fiber_switch expects it: six callee-saved registers, then the address it returns to; the
        // trampoline starts with rsp = top - 8 (as after a call from a 16-byte aligned frame) and never returns
        void** top = (void**)(((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15);
        top[-1] = nullptr;
        top[-2] = (void*)&trampoline;
        for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
        f.sp = (void*)(top - 8);
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
#endif
    }
    for (;;) {
        bool progressed = false;
        for (int k = 0; k < n; ++k) {
            Fiber& f = g_fibers[reverse ? n - 1 - k : k];
            if (f.state == READY) {
                g_cur = &f;
#if SIMT_FAST_SWITCH
                fiber_switch(&g_sched_sp, f.sp);
#else
                swapcontext(&g_sched, &f.ctx);
#endif
                progressed = true;
            }
        }
        int n_bar = 0, n_done = 0;
        for (int t = 0; t < n; ++t) {
            n_bar += g_fibers[t].state == AT_BARRIER;
            n_done += g_fibers[t].state == DONE;
        }
        if (n_done == n) break;
        if (n_bar > 0 && n_bar + n_done == n) {
            for (int t = 0; t < n; ++t)
                if (g_fibers[t].state == AT_BARRIER) g_fibers[t].state = READY;
            progressed = true;
        }
        if (!progressed) die("deadlock: divergent barrier / wave collective");
    }
    g_cur = nullptr;
}

template <class F>
inline void launch(dim3 grid, dim3 block, F&& f) {
    std::function<void()> body = f;
    g_gridDim = grid;
    g_blockDim = block;
    g_nthreads = (int)(block.x * block.y * block.z);
    if (g_nthreads <= 0 || g_nthreads > 1024) die("bad block size");
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = simt_uint3{bx, by, bz};
                run_block(body);
            }
}

inline int lane_id() { return g_cur->linear % WAVE; }
inline WaveSync& my_wave() { return g_waves[g_cur->linear / WAVE]; }

template <class T>
inline T wave_read(T mine, int src_lane) {
    static_assert(sizeof(T) <= XCHG_BYTES, "exchange slot too small");
    WaveSync& ws = my_wave();
    memcpy(ws.xchg[lane_id()], &mine, sizeof(T));
    wave_sync();
    T r;
    memcpy(&r, ws.xchg[src_lane & (WAVE - 1)], sizeof(T));
    wave_sync();
    return r;
}

}  // namespace simt

#define threadIdx (simt::g_cur->tid)
#define blockIdx (simt::g_blockIdx)
#define blockDim (simt::g_blockDim)
#define gridDim (simt::g_gridDim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (simt::set_dynamic_smem(shmem), simt::launch((grid), (block), [=]() { kernel(__VA_ARGS__); }))

// dynamic shared memory: one 16-byte aligned host buffer per launch
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)simt::g_dyn_smem;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <class F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

inline void __syncthreads() { simt::block_barrier(); }

template <class T> inline T __shfl(T v, int src, int width = 64) {
    int l = simt::lane_id();
    int base = l & ~(width - 1);
    return simt::wave_read(v, base + (src & (width - 1)));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = simt::lane_id();
    int t = l ^ mask;
    if ((t & ~(width - 1)) != (l & ~(width - 1))) t = l;
    return simt::wave_read(v, t);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = simt::lane_id();
    int t = l + (int)d;
    if ((t & ~(width - 1)) != (l & ~(width - 1))) t = l;
    return simt::wave_read(v, t);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = simt::lane_id();
    int t = l - (int)d;
    if (t < 0 || (t & ~(width - 1)) != (l & ~(width - 1))) t = l;
    return simt::wave_read(v, t);
}
inline unsigned long long __ballot(int pred) {      // one exchange: every lane posts its predicate, then reads all slots
    simt::WaveSync& ws = simt::my_wave();
    memcpy(ws.xchg[simt::lane_id()], &pred, sizeof pred);
    simt::wave_sync();
    unsigned long long m = 0;
    const int lanes = simt::wave_lanes(simt::g_cur->linear / simt::WAVE);
    for (int s = 0; s < lanes; ++s) {
        int p;
        memcpy(&p, ws.xchg[s], sizeof p);
        if (p) m |= 1ull << s;
    }
    simt::wave_sync();
    return m;
}
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __builtin_amdgcn_readlane(int v, int lane) { return simt::wave_read(v, lane); }     // v_readlane_b32
// v_perm_b32: byte i of the result = byte sel[i] of the 8 bytes {hi : lo} (0..3 = lo, 4..7 = hi); selectors >= 8 are
// not used by the kernels (constants: 0x0c = 0x00)
inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel) {
    const unsigned long long src = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned s = (sel >> (8 * i)) & 0xFFu;
        const unsigned b = s < 8 ? (unsigned)((src >> (8 * s)) & 0xFFu) : (s == 0x0c ? 0u : 0xFFu);
        r |= b << (8 * i);
    }
    return r;
}
inline int __builtin_amdgcn_readfirstlane(int v) { return simt::wave_read(v, 0); }              // v_readfirstlane_b32 (all lanes active)
// DPP lane exchanges used by the kernels: quad_perm (ctrl < 0x100), row_mirror (0x140), row_half_mirror (0x141),
// row_newbcast:k (0x150 + k)
inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool) {
    const int l = simt::lane_id();
    int t;
    if (ctrl < 0x100) t = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl == 0x140) t = (l & ~15) | (15 - (l & 15));
    else if (ctrl == 0x141) t = (l & ~7) | (7 - (l & 7));
    else if (ctrl >= 0x150 && ctrl <= 0x15F) t = (l & ~15) | (ctrl - 0x150);
    else abort();
    return simt::wave_read(src, t);
}
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
// LDS-DMA: destination = the FIRST lane's LDS pointer + lane * size (wave-linear), source per lane
template <class PS, class PD>
inline void __builtin_amdgcn_global_load_lds(PS src, PD dst, unsigned size, int offset, unsigned) {
    unsigned char* base = (unsigned char*)simt::wave_read((uintptr_t)dst, 0);
    memcpy(base + (size_t)simt::lane_id() * size, (const unsigned char*)(uintptr_t)src + offset, size);
    simt::wave_sync();
}
// ds_read_b64_tr_b16 (gfx950; lane map measured on the hardware by tools/probe/tr_probe.hip): every lane
// addresses 4 consecutive halfwords; lane i of a 16-lane group, element j <- the halfword that lane
// 4*j + (i >> 2) of the same group addresses, its element i & 3 (a [4][16] block leaves transposed).
typedef short simt_s16x4 __attribute__((ext_vector_type(4)));
template <class P>
inline simt_s16x4 __builtin_amdgcn_ds_read_tr16_b64_v4i16(P ptr) {
    if (((uintptr_t)ptr) & 7) simt::die("ds_read_b64_tr_b16: address not 8-byte aligned");
    simt_s16x4 mine;
    memcpy(&mine, (const void*)(uintptr_t)ptr, sizeof mine);
    simt::WaveSync& ws = simt::my_wave();
    const int l = simt::lane_id();
    memcpy(ws.xchg[l], &mine, sizeof mine);
    simt::wave_sync();
    simt_s16x4 r;
    const int grp = l & ~15, i = l & 15;
    for (int j = 0; j < 4; ++j) {
        simt_s16x4 other;
        memcpy(&other, ws.xchg[grp + 4 * j + (i >> 2)], sizeof other);
        r[j] = other[i & 3];
    }
    simt::wave_sync();
    return r;
}
inline void __builtin_amdgcn_fence(int, const char*) {}
inline void __builtin_amdgcn_wave_barrier() { simt::wave_sync(); }
inline void __builtin_amdgcn_s_barrier() { simt::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

// ---- vector types used by the MFMA builtins (clang ext vectors work on the host too)
typedef float simt_f32x4 __attribute__((ext_vector_type(4)));
typedef float simt_f32x16 __attribute__((ext_vector_type(16)));
typedef double simt_f64x4 __attribute__((ext_vector_type(4)));
typedef short simt_s16x8 __attribute__((ext_vector_type(8)));

inline float simt_bf16_to_f32(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline simt_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, simt_f32x16 c, int, int, int) {
    struct AB { float a, b; } mine{a, b};
    simt::WaveSync& ws = simt::my_wave();
    int l = simt::lane_id();
    memcpy(ws.xchg[l], &mine, sizeof mine);
    simt::wave_sync();
    simt_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            AB pa, pb;
            memcpy(&pa, ws.xchg[row + 32 * k], sizeof pa);
            memcpy(&pb, ws.xchg[col + 32 * k], sizeof pb);
            acc = fmaf(pa.a, pb.b, acc);
        }
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}

inline simt_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, simt_f32x4 c, int, int, int) {
    struct AB { float a, b; } mine{a, b};
    simt::WaveSync& ws = simt::my_wave();
    int l = simt::lane_id();
    memcpy(ws.xchg[l], &mine, sizeof mine);
    simt::wave_sync();
    simt_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            AB pa, pb;
            memcpy(&pa, ws.xchg[row + 16 * k], sizeof pa);
            memcpy(&pb, ws.xchg[col + 16 * k], sizeof pb);
            acc = fmaf(pa.a, pb.b, acc);
        }
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}

inline simt_f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, simt_f64x4 c, int, int, int) {
    struct AB { double a, b; } mine{a, b};
    simt::WaveSync& ws = simt::my_wave();
    int l = simt::lane_id();
    memcpy(ws.xchg[l], &mine, sizeof mine);
    simt::wave_sync();
    simt_f64x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) + 4 * r, col = l & 15;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) {
            AB pa, pb;
            memcpy(&pa, ws.xchg[row + 16 * k], sizeof pa);
            memcpy(&pb, ws.xchg[col + 16 * k], sizeof pb);
            acc = fma(pa.a, pb.b, acc);
        }
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}

typedef __bf16 simt_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short simt_u16x2 __attribute__((ext_vector_type(2)));
inline float simt_bf16_to_f32(unsigned short h);
inline float __builtin_amdgcn_fdot2_f32_bf16(simt_bf16x2 a, simt_bf16x2 b, float c, bool) {      // v_dot2c_f32_bf16
    const simt_u16x2 ua = __builtin_bit_cast(simt_u16x2, a), ub = __builtin_bit_cast(simt_u16x2, b);
    return c + simt_bf16_to_f32(ua[0]) * simt_bf16_to_f32(ub[0]) + simt_bf16_to_f32(ua[1]) * simt_bf16_to_f32(ub[1]);
}
typedef __bf16 simt_bf16x8 __attribute__((ext_vector_type(8)));
inline simt_f32x4 simt_mfma_f32_16x16x32_bf16(simt_s16x8 a, simt_s16x8 b, simt_f32x4 c);
inline simt_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(simt_bf16x8 a, simt_bf16x8 b, simt_f32x4 c, int, int, int) {
    return simt_mfma_f32_16x16x32_bf16(__builtin_bit_cast(simt_s16x8, a), __builtin_bit_cast(simt_s16x8, b), c);
}

inline simt_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(simt_bf16x8 a, simt_bf16x8 b, simt_f32x16 c, int, int, int) {
    struct AB { simt_s16x8 a, b; } mine{__builtin_bit_cast(simt_s16x8, a), __builtin_bit_cast(simt_s16x8, b)};
    simt::WaveSync& ws = simt::my_wave();
    int l = simt::lane_id();
    memcpy(ws.xchg[l], &mine, sizeof mine);
    simt::wave_sync();
    simt_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            AB pa, pb;
            memcpy(&pa, ws.xchg[row + 32 * (k >> 3)], sizeof pa);
            memcpy(&pb, ws.xchg[col + 32 * (k >> 3)], sizeof pb);
            acc += simt_bf16_to_f32((unsigned short)pa.a[k & 7]) * simt_bf16_to_f32((unsigned short)pb.b[k & 7]);
        }
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}

// bf16 operands are passed as 8 raw 16-bit patterns per lane
inline simt_f32x4 simt_mfma_f32_16x16x32_bf16(simt_s16x8 a, simt_s16x8 b, simt_f32x4 c) {
    struct AB { simt_s16x8 a, b; } mine{a, b};
    simt::WaveSync& ws = simt::my_wave();
    int l = simt::lane_id();
    memcpy(ws.xchg[l], &mine, sizeof mine);
    simt::wave_sync();
    simt_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            AB pa, pb;
            memcpy(&pa, ws.xchg[row + 16 * (k >> 3)], sizeof pa);
            memcpy(&pb, ws.xchg[col + 16 * (k >> 3)], sizeof pb);
            acc += simt_bf16_to_f32((unsigned short)pa.a[k & 7]) * simt_bf16_to_f32((unsigned short)pb.b[k & 7]);
        }
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}
