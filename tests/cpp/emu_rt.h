// emu_rt.h — a CPU stand-in for the parts of HIP that libptcore uses, so that the library's OWN sources (core.hip, trace.hip,
// trace_noalloc.hip, render.hip, tonemap.hip, lbvh.hip: host logic and kernels alike) compile as ordinary C++ and run on the host.  TEST INFRASTRUCTURE:
// force-included (-include) in front of those sources by tests/cpp/Makefile to build tests/cpp/libptcore_emu.so, which the
// CPU suite loads through the same ctypes binding as the real library (tests/test_emu_lib.py), optionally under the host's
// address sanitizer.  Nothing here is part of the product.
//
// Kernels: hipLaunchKernelGGL runs the grid at once, block after block; every lane of a block is a fiber (ucontext); a wave's
// lanes run one after the other up to their next collective (__ballot, __shfl, readfirstlane, __syncthreads); the scheduler
// checks that the wave's live lanes all stopped at the same collective, resolves it and picks the next wave of the block at
// random (seed: PTC_EMU_SEED), and with PTC_EMU_SHUFFLE=1 the blocks of every launch run in a random order.  Atomics are plain read-modify-writes (a stretch between two collectives is atomic here),
// __shared__ arrays are statics (one block runs at a time).  Streams and events do nothing: every call completes before it
// returns, so the host logic that runs ahead of the device (render.hip) sees its counters two iterations "late" but correct.
// What this does not model: memory ordering inside a stretch, concurrency between blocks, timing.
#ifndef PTC_EMU_RT_H
#define PTC_EMU_RT_H

#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif

#define PTC_EMU 1

namespace emu {

struct Dim { uint32_t x, y, z; };
enum Kind { K_NONE = 0, K_BALLOT, K_FIRST, K_SHFL, K_BARRIER };

struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    Dim tid{0, 0, 0};
    bool done = true;
    int kind = K_NONE, site = 0;
    uint64_t arg = 0, arg2 = 0, result = 0;
};

inline Fiber* cur = nullptr;            // the lane that is running
inline ucontext_t sched_ctx;
inline Dim g_block{0, 1, 1}, g_grid{1, 1, 1}, g_bdim{256, 1, 1};
inline std::function<void()> g_entry;
inline uint64_t g_switches = 0, g_launches = 0;
inline const char* g_error = nullptr;   // sticky, like a HIP launch failure: hipGetLastError reports and clears it
inline std::mt19937_64 g_rng{[] { const char* s = getenv("PTC_EMU_SEED"); return s ? strtoull(s, nullptr, 10) : 1ull; }()};
inline const void* sched_stack = nullptr;      // the scheduler's (= the calling thread's) stack, as the sanitizer reports it
inline size_t sched_stack_size = 0;
inline std::vector<Fiber> g_fibers;

inline void to_scheduler() {
    Fiber* const me = cur;
#ifdef EMU_ASAN
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, sched_stack, sched_stack_size);
#endif
    swapcontext(&me->ctx, &sched_ctx);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(fake, &sched_stack, &sched_stack_size);
#endif
}

inline uint64_t collective(int kind, uint64_t arg, int site, uint64_t arg2 = 0) {
    cur->kind = kind; cur->arg = arg; cur->arg2 = arg2; cur->site = site;
    to_scheduler();
    return cur->result;
}

inline void trampoline() {
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &sched_stack, &sched_stack_size);
#endif
    g_entry();
    cur->done = true;
    cur->kind = K_NONE;
    cur->site = 0;
#ifdef EMU_ASAN
    __sanitizer_start_switch_fiber(nullptr, sched_stack, sched_stack_size);    // the fiber ends: its fake stack may go
#endif
    swapcontext(&cur->ctx, &sched_ctx);
}

inline void resume(Fiber& f) {
    cur = &f;
    g_switches += 1;
#ifdef EMU_ASAN
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, f.stack.data(), f.stack.size());
#endif
    swapcontext(&sched_ctx, &f.ctx);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
    cur = nullptr;
}

// one block of g_entry with `threads` lanes; false (and g_error) when a wave's lanes disagree on a collective or
// `max_steps` wave stretches did not finish it
inline bool run_block(uint32_t block, uint32_t grid, uint32_t threads, uint64_t max_steps) {
    g_block = {block, 0, 0};
    g_grid = {grid, 1, 1};
    g_bdim = {threads, 1, 1};
    const uint32_t waves = (threads + 63u) / 64u;
    if (g_fibers.size() < threads) {
        g_fibers.resize(threads);
        for (auto& f : g_fibers) if (f.stack.empty()) f.stack.resize(512 * 1024);
    }
    for (uint32_t t = 0; t < threads; ++t) {
        Fiber& f = g_fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.data();
        f.ctx.uc_stack.ss_size = f.stack.size();
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, trampoline, 0);
        f.tid = {t, 0, 0};
        f.done = false;
        f.site = 0;
        f.kind = K_NONE;
    }
    std::vector<char> at_barrier(waves, 0);
    std::vector<uint32_t> live(waves);
    for (uint64_t step = 0; step < max_steps; ++step) {
        // waves that can run: lanes left, and not held at a barrier
        uint32_t n_live = 0, n_alive = 0, n_waiting = 0;
        for (uint32_t w = 0; w < waves; ++w) {
            bool any = false;
            for (uint32_t t = w * 64; t < threads && t < w * 64 + 64; ++t) any = any || !g_fibers[t].done;
            if (!any) continue;
            n_alive += 1;
            if (at_barrier[w]) { n_waiting += 1; continue; }
            live[n_live++] = w;
        }
        if (n_alive == 0) return true;
        if (n_live == 0) {                       // every wave that is left waits at the barrier: release them
            if (n_waiting != n_alive) { g_error = "barrier bookkeeping"; return false; }
            for (uint32_t w = 0; w < waves; ++w) at_barrier[w] = 0;
            continue;
        }
        const uint32_t w = live[g_rng() % n_live];
        Fiber* lanes = &g_fibers[w * 64];
        const uint32_t nl = threads - w * 64 < 64u ? threads - w * 64 : 64u;
        // resolve the collective the wave's lanes stopped at
        int kind = K_NONE, site = 0;
        uint64_t mask = 0, first = 0;
        bool have_first = false;
        for (uint32_t l = 0; l < nl; ++l) {
            if (lanes[l].done) continue;
            if (kind == K_NONE && lanes[l].kind != K_NONE) { kind = lanes[l].kind; site = lanes[l].site; }
            if (lanes[l].kind != kind || lanes[l].site != site) {
                std::fprintf(stderr, "emu: wave %u lane %u at kind %d line %d, an earlier lane at kind %d line %d\n", w, l, lanes[l].kind, lanes[l].site, kind, site);
                g_error = "lanes of a wave stopped at different collectives";
                return false;
            }
            if (lanes[l].arg) mask |= 1ull << l;
            if (!have_first) { first = lanes[l].arg; have_first = true; }
        }
        for (uint32_t l = 0; l < nl; ++l) {
            if (lanes[l].done) continue;
            if (kind == K_SHFL) {                // the value of the named lane (an inactive source: the lane's own, as undefined as on the GPU)
                const uint32_t src = (uint32_t)lanes[l].arg2 & 63u;
                lanes[l].result = (src < nl && !lanes[src].done) ? lanes[src].arg : lanes[l].arg;
            } else {
                lanes[l].result = kind == K_BALLOT ? mask : first;
            }
        }
        // run every lane to its next collective (or its end)
        for (uint32_t l = 0; l < nl; ++l) if (!lanes[l].done) resume(lanes[l]);
        bool barrier = false, other = false;
        for (uint32_t l = 0; l < nl; ++l) {
            if (lanes[l].done) continue;
            if (lanes[l].kind == K_BARRIER) barrier = true; else other = true;
        }
        if (barrier && other) { g_error = "some lanes of a wave at a barrier, others not"; return false; }
        at_barrier[w] = barrier;
    }
    g_error = "the block did not finish";
    return false;
}

// a whole grid, block after block (a valid schedule of any grid whose blocks do not wait for each other)
template <class F>
inline void launch(uint32_t grid, uint32_t threads, F&& body) {
    g_launches += 1;
    if (g_error) return;                   // like a sticky HIP error: nothing runs after a failed launch until it was read
    if (grid == 0 || threads == 0 || threads > 1024) { g_error = "bad launch configuration"; return; }
    g_entry = std::forward<F>(body);
    // PTC_EMU_SHUFFLE=1: the blocks of a launch run in a random order (the GPU's dispatch order is undefined: nothing may depend on it)
    static const bool shuffle = [] { const char* v = getenv("PTC_EMU_SHUFFLE"); return v && atoi(v) != 0; }();
    std::vector<uint32_t> order;
    if (shuffle) {
        order.resize(grid);
        for (uint32_t b = 0; b < grid; ++b) order[b] = b;
        for (uint32_t b = grid; b > 1; --b) std::swap(order[b - 1], order[g_rng() % b]);
    }
    for (uint32_t i = 0; i < grid; ++i) {
        const uint32_t b = shuffle ? order[i] : i;
        if (!run_block(b, grid, threads, 1ull << 28)) {
            std::fprintf(stderr, "emu: block %u of %u: %s\n", b, grid, g_error ? g_error : "?");
            return;
        }
    }
}

}  // namespace emu

// ---- the device vocabulary the kernels use ------------------------------------------------------------------------------
#undef __shared__
#define __shared__ static
#undef __global__
#define __global__
#undef __device__
#define __device__
#undef __host__
#define __host__
#undef __forceinline__
#define __forceinline__ inline __attribute__((always_inline))
#undef __launch_bounds__
#define __launch_bounds__(...)
#define threadIdx (emu::cur->tid)
#define blockIdx (emu::g_block)
#define gridDim (emu::g_grid)
#define blockDim (emu::g_bdim)
#define __ballot(p) emu::collective(emu::K_BALLOT, (p) ? 1u : 0u, __LINE__)
// (int -> int like the builtin, so that a 64-bit value put together from two of them without care goes wrong here too)
#define __builtin_amdgcn_readfirstlane(v) ((int)emu::collective(emu::K_FIRST, (uint64_t)(uint32_t)(v), __LINE__))
#define __shfl(v, lane) ((uint32_t)emu::collective(emu::K_SHFL, (uint64_t)(uint32_t)(v), __LINE__, (uint64_t)(lane)))
#define __syncthreads() ((void)emu::collective(emu::K_BARRIER, 0, __LINE__))
// (the value of lane (own ^ mask), bits preserved for any 4-byte type)
template <class T> static inline T emu_shfl_xor(T v, int mask, int site) {
    static_assert(sizeof(T) == 4, "4-byte values only");
    uint32_t bits; std::memcpy(&bits, &v, 4);
    const uint32_t r = (uint32_t)emu::collective(emu::K_SHFL, bits, site, (uint64_t)((emu::cur->tid.x & 63u) ^ (uint32_t)mask));
    T o; std::memcpy(&o, &r, 4); return o;
}
#define __shfl_xor(v, mask) emu_shfl_xor((v), (mask), __LINE__)
#define __clz(x) ((x) ? __builtin_clz((unsigned)(x)) : 32)
#define __clzll(x) ((x) ? __builtin_clzll((unsigned long long)(x)) : 64)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
#define __ffsll(x) __builtin_ffsll(x)
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
// (a stretch between two collectives is atomic, so plain read-modify-writes are atomic enough)
template <typename T, typename U> static inline T atomicAdd(T* p, U v) { const T o = *p; *p = o + (T)v; return o; }
template <typename T, typename U> static inline T atomicSub(T* p, U v) { const T o = *p; *p = o - (T)v; return o; }
template <typename T, typename U> static inline T atomicOr(T* p, U v) { const T o = *p; *p = o | (T)v; return o; }
template <typename T, typename U> static inline T atomicAnd(T* p, U v) { const T o = *p; *p = o & (T)v; return o; }
template <typename T, typename U> static inline T atomicMax(T* p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicMin(T* p, U v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename U, typename V> static inline T atomicCAS(T* p, U expect, V v) { const T o = *p; if (o == (T)expect) *p = (T)v; return o; }

#pragma clang diagnostic ignored "-Wunknown-attributes"
#pragma clang diagnostic ignored "-Wignored-attributes"

// ---- the host API: memory is the host's, everything is synchronous ------------------------------------------------------
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch((uint32_t)dim3(grid).x, (uint32_t)dim3(block).x, [=] { kern(__VA_ARGS__); })

namespace emu {
inline hipError_t ok() { return hipSuccess; }
inline hipError_t last_error() { if (g_error) { g_error = nullptr; return hipErrorLaunchFailure; } return hipSuccess; }
inline hipError_t alloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t malloc_t(T** p, size_t n) { return alloc((void**)p, n); }
inline hipError_t release(void* p) { std::free(p); return hipSuccess; }
inline hipError_t copy(void* d, const void* s, size_t n) { if (n) std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t set(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
template <class T> inline hipError_t handle(T* h) { *h = nullptr; return hipSuccess; }
inline hipError_t props(hipDeviceProp_t* p) {
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "ptcore CPU emulation");
    const char* s = getenv("PTC_EMU_CUS");
    p->multiProcessorCount = s ? atoi(s) : 8;      // small grids: a CU is five blocks of fibers here
    p->totalGlobalMem = (size_t)8 << 30;
    return hipSuccess;
}
}  // namespace emu

#undef hipGetDeviceProperties
#define hipGetDeviceCount(n) (*(n) = 1, hipSuccess)
#define hipSetDevice(d) emu::ok()
#define hipGetDeviceProperties(p, d) emu::props(p)
#define hipGetLastError() emu::last_error()
#define hipGetErrorString(e) ((e) == hipSuccess ? "no error" : (e) == hipErrorOutOfMemory ? "out of memory (emulation)" : "a kernel failed in the CPU emulation (see stderr)")
#define hipMalloc(p, n) emu::malloc_t((p), (n))
#define hipFree(p) emu::release((void*)(p))
#define hipHostMalloc(p, n, flags) emu::malloc_t((p), (n))
#define hipHostFree(p) emu::release((void*)(p))
#define hipPointerGetAttributes(a, p) ((a)->type = hipMemoryTypeUnregistered, hipSuccess)
#define hipMemcpy(d, s, n, kind) emu::copy((d), (s), (n))
#define hipMemcpyAsync(d, s, n, kind, stream) emu::copy((d), (s), (n))
#define hipMemset(d, v, n) emu::set((d), (v), (n))
#define hipMemsetAsync(d, v, n, stream) emu::set((d), (v), (n))
#define hipStreamCreate(s) emu::handle(s)
#define hipStreamCreateWithFlags(s, f) emu::handle(s)
#define hipStreamDestroy(s) emu::ok()
#define hipStreamSynchronize(s) emu::last_error()
#define hipStreamWaitEvent(s, e, f) emu::ok()
#define hipEventCreate(e) emu::handle(e)
#define hipEventCreateWithFlags(e, f) emu::handle(e)
#define hipEventDestroy(e) emu::ok()
#define hipEventRecord(e, s) emu::ok()
#define hipEventSynchronize(e) emu::last_error()
#define hipEventElapsedTime(ms, a, b) (*(ms) = 0.0f, hipSuccess)
// (trace_noalloc.hip: its file-scope `static __device__` scratch array is a host array here, and its address is its address)
#undef HIP_SYMBOL
#define HIP_SYMBOL(x) (x)
#define hipGetSymbolAddress(pp, sym) (*(pp) = (void*)(sym), hipSuccess)
#undef __align__
#define __align__(n) __attribute__((aligned(n)))

#endif
