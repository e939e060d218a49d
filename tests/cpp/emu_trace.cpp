// emu_trace.cpp — the traversal kernels' SOURCE (pathtracer_amd/csrc/trace_kernels.h: k_trace_scoped_persistent,
// k_trace_replay_persistent, k_trace_mint_persistent) compiled for the CPU and run lane by lane.  TEST INFRASTRUCTURE: built
// into tests/cpp/libemutrace.so (and, with the host's address sanitizer, libemutrace_asan.so), loaded by
// tests/test_emu_trace.py only.
//
// Why: what the CPU suite could check of the kernels so far was their shared per-record decision code (trace_step.h, through
// tests/cpp/scoped_model.cpp); the ring stack over the spill column, the ray tickets, the refill, the redo list and the frame
// replay existed on the GPU only, where a mistake is a memory fault.  Here every lane of a block is a fiber (ucontext); a
// wave's lanes run one after the other up to their next wave collective (__ballot, readfirstlane, __syncthreads), the
// scheduler resolves the collective and picks the next wave AT RANDOM (seeded), every LDS and global access is
// bounds-checked by the sanitizer (the buffers have the launcher's exact sizes), and the hits are compared with the oracle's
// bit for bit.  What this is not: a model of memory ordering inside a wave's stretch between two collectives (a stretch
// is atomic here), or of timing.  (It was built to find a fault in a round-3 experiment, tools/experiments/
// r03_cross_wave_drain.diff.txt, whose cross-wave protocol it ran under random schedules; that experiment was not kept,
// this was.)
//
// Blocks run one after the other (a valid schedule of a grid whose blocks only meet at the ticket words).
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "../../pathtracer_amd/csrc/device_types.h"
#include "../../pathtracer_amd/csrc/trace_step.h"

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif

namespace emu {

struct Dim { uint32_t x, y, z; };
enum Kind { K_NONE = 0, K_BALLOT, K_FIRST, K_BARRIER };

struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    Dim tid{0, 0, 0};
    bool done = true;
    int kind = K_NONE, site = 0;
    uint64_t arg = 0, result = 0;
};

Fiber* cur = nullptr;            // the lane that is running
ucontext_t sched_ctx;
Dim g_block{0, 1, 1}, g_grid{1, 1, 1};
void (*g_entry)() = nullptr;
uint64_t g_switches = 0;
const char* g_error = nullptr;

const void* sched_stack = nullptr;      // the scheduler's (= the calling thread's) stack, as the sanitizer reports it
size_t sched_stack_size = 0;

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

inline uint64_t collective(int kind, uint64_t arg, int site) {
    cur->kind = kind; cur->arg = arg; cur->site = site;
    to_scheduler();
    return cur->result;
}

void trampoline() {
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

void resume(Fiber& f) {
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

constexpr uint32_t BLOCK = 256, WAVES = BLOCK / 64;
std::vector<Fiber> g_fibers;

// one block of `entry`, waves scheduled by `rng`; false (and g_error) when the block's lanes disagree on a collective or
// `max_steps` wave stretches did not finish it
bool run_block(void (*entry)(), uint32_t block, uint32_t grid, std::mt19937_64& rng, uint64_t max_steps) {
    g_entry = entry;
    g_block = {block, 0, 0};
    g_grid = {grid, 1, 1};
    if (g_fibers.size() != BLOCK) {
        g_fibers.resize(BLOCK);
        for (auto& f : g_fibers) f.stack.resize(512 * 1024);
    }
    for (uint32_t t = 0; t < BLOCK; ++t) {
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
    bool at_barrier[WAVES] = {false, false, false, false};
    for (uint64_t step = 0; step < max_steps; ++step) {
        // waves that can run: lanes left, and not held at a barrier
        uint32_t live[WAVES], n_live = 0, n_alive = 0, n_waiting = 0;
        for (uint32_t w = 0; w < WAVES; ++w) {
            bool any = false;
            for (uint32_t l = 0; l < 64; ++l) any = any || !g_fibers[w * 64 + l].done;
            if (!any) continue;
            n_alive += 1;
            if (at_barrier[w]) { n_waiting += 1; continue; }
            live[n_live++] = w;
        }
        if (n_alive == 0) return true;
        if (n_live == 0) {                       // every wave that is left waits at the barrier: release them
            if (n_waiting != n_alive) { g_error = "barrier bookkeeping"; return false; }
            for (uint32_t w = 0; w < WAVES; ++w) at_barrier[w] = false;
            continue;
        }
        const uint32_t w = live[rng() % n_live];
        Fiber* lanes = &g_fibers[w * 64];
        // resolve the collective the wave's lanes stopped at
        int kind = K_NONE, site = 0;
        uint64_t mask = 0, first = 0;
        bool have_first = false;
        for (uint32_t l = 0; l < 64; ++l) {
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
        for (uint32_t l = 0; l < 64; ++l) lanes[l].result = kind == K_BALLOT ? mask : first;
        // run every lane to its next collective (or its end)
        for (uint32_t l = 0; l < 64; ++l) if (!lanes[l].done) resume(lanes[l]);
        bool barrier = false, other = false;
        for (uint32_t l = 0; l < 64; ++l) {
            if (lanes[l].done) continue;
            if (lanes[l].kind == K_BARRIER) barrier = true; else other = true;
        }
        if (barrier && other) { g_error = "some lanes of a wave at a barrier, others not"; return false; }
        at_barrier[w] = barrier;
    }
    g_error = "the block did not finish";
    return false;
}

}  // namespace emu

// ---- the device vocabulary the kernels use ------------------------------------------------------------------------------
#undef __shared__
#define __shared__ static
#undef __global__
#define __global__
#undef __device__
#define __device__
#undef __forceinline__
#define __forceinline__ inline __attribute__((always_inline))
#undef __launch_bounds__
#define __launch_bounds__(...)
#define threadIdx (emu::cur->tid)
#define blockIdx (emu::g_block)
#define gridDim (emu::g_grid)
#define __ballot(p) emu::collective(emu::K_BALLOT, (p) ? 1u : 0u, __LINE__)
// (int -> int like the builtin, so that a 64-bit value put together from two of them without care goes wrong here too)
#define __builtin_amdgcn_readfirstlane(v) ((int)emu::collective(emu::K_FIRST, (uint64_t)(uint32_t)(v), __LINE__))
#define __syncthreads() ((void)emu::collective(emu::K_BARRIER, 0, __LINE__))
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
// (a stretch between two collectives is atomic, so plain read-modify-writes are atomic enough)
template <typename T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicSub(T* p, T v) { const T o = *p; *p = o - v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T* p, T v) { const T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicCAS(T* p, T expect, T v) { const T o = *p; if (o == expect) *p = v; return o; }

#define PTC_EMU 1
#pragma clang diagnostic ignored "-Wunknown-attributes"
#pragma clang diagnostic ignored "-Wignored-attributes"
#include "../../pathtracer_amd/csrc/trace_kernels.h"

namespace {

inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// Triangle::new (triangle.rs:50-60) into the kernel's [n_tris][PTC_TRI_STRIDE] records
void precompute(const pt_triangle& t, float4* rec) {
    float ab[3], ac[3], n[3], n1[3], n2[3], tmp[3];
    for (int k = 0; k < 3; ++k) { ab[k] = t.b[k] - t.a[k]; ac[k] = t.c[k] - t.a[k]; }
    cross3(ab, ac, n);
    const float dd = dot3(t.a, n), nn = dot3(n, n);
    cross3(ac, n, tmp);
    for (int k = 0; k < 3; ++k) n1[k] = tmp[k] / nn;
    const float d1 = -dot3(t.a, n1);
    cross3(n, ab, tmp);
    for (int k = 0; k < 3; ++k) n2[k] = tmp[k] / nn;
    const float d2 = -dot3(t.a, n2);
    rec[0] = make_float4(n[0], n[1], n[2], dd);
    rec[1] = make_float4(n1[0], n1[1], n1[2], d1);
    rec[2] = make_float4(n2[0], n2[1], n2[2], d2);
    rec[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// the wide record of binary node i, indexed by the node's own index (device_types.h: slots 0,1 = right child's right / left,
// 2,3 = left child's right / left; a leaf child in the first slot of its pair, NaN box in the other)
void make_wide(const pt_bvh_node* nodes, uint32_t i, float4* w) {
    const float nan = std::numeric_limits<float>::quiet_NaN();
    float lo[3][4], hi[3][4];
    uint32_t ref[4] = {PTC_IDLE, PTC_IDLE, PTC_IDLE, PTC_IDLE};
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 4; ++k) { lo[a][k] = nan; hi[a][k] = nan; }
    auto put = [&](int slot, const pt_aabb& b, uint32_t child) {
        for (int a = 0; a < 3; ++a) { lo[a][slot] = b.min[a]; hi[a][slot] = b.max[a]; }
        ref[slot] = nodes[child].is_leaf ? (PTC_LEAF_BIT | nodes[child].shape_index) : child;
    };
    const pt_bvh_node& x = nodes[i];
    for (int side = 1; side >= 0; --side) {
        const uint32_t c = side ? x.child_r_index : x.child_l_index;
        const int base = side ? 0 : 2;
        if (nodes[c].is_leaf) put(base, side ? x.child_r_aabb : x.child_l_aabb, c);
        else { put(base, nodes[c].child_r_aabb, nodes[c].child_r_index); put(base + 1, nodes[c].child_l_aabb, nodes[c].child_l_index); }
    }
    for (int a = 0; a < 3; ++a) {
        w[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
        w[3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
    }
    w[6] = make_float4(u2f(ref[0]), u2f(ref[1]), u2f(ref[2]), u2f(ref[3]));
    w[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

uint32_t depth_of(const pt_bvh_node* nodes, uint32_t n_nodes) {     // nodes on the longest root-to-leaf path (iterative)
    if (n_nodes == 0) return 0;
    std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 1u}};
    uint32_t best = 0;
    while (!todo.empty()) {
        auto [i, d] = todo.back();
        todo.pop_back();
        best = d > best ? d : best;
        if (!nodes[i].is_leaf) { todo.push_back({nodes[i].child_l_index, d + 1}); todo.push_back({nodes[i].child_r_index, d + 1}); }
    }
    return best;
}

// the device layout of the binary nodes (device_types.h), indexed by the node's own index (leaves leave a gap)
void make_node(const pt_bvh_node* nodes, uint32_t i, float4* q) {
    const pt_bvh_node& x = nodes[i];
    auto ref = [&](uint32_t c) { return nodes[c].is_leaf ? (PTC_LEAF_BIT | nodes[c].shape_index) : c; };
    q[0] = make_float4(x.child_l_aabb.min[0], x.child_l_aabb.min[1], x.child_l_aabb.min[2], u2f(ref(x.child_l_index)));
    q[1] = make_float4(x.child_l_aabb.max[0], x.child_l_aabb.max[1], x.child_l_aabb.max[2], u2f(ref(x.child_r_index)));
    q[2] = make_float4(x.child_r_aabb.min[0], x.child_r_aabb.min[1], x.child_r_aabb.min[2], 0.0f);
    q[3] = make_float4(x.child_r_aabb.max[0], x.child_r_aabb.max[1], x.child_r_aabb.max[2], 0.0f);
}

// the arguments of the launch that is running (the kernels take them by value; the fibers' entry has none)
struct Launch {
    DevSceneView sc;
    const float4* rays; const uint32_t* start; const uint32_t* queue; const uint32_t* count_ptr; uint32_t n;
    uint4* out; uint2* spill; uint32_t total_threads;
    uint32_t *ticket, *ticket_next, refill, ticket_size, *redo_list, *redo_count, *redo_count_next;
} g_launch;

uint32_t g_ring = 12;      // LDS ring slots per lane of the scoped kernel (PTC_RING: 8, 10, 12 or 16)
template <int S>
void scoped() {
    const Launch& a = g_launch;
    k_trace_scoped_persistent<S, false>(a.sc, a.rays, a.start, nullptr, nullptr, a.n, a.out, a.spill, a.total_threads, a.ticket,
                                        a.ticket_next, a.refill, a.ticket_size, a.redo_list, a.redo_count, a.redo_count_next);
}
void entry_scoped() {
    switch (g_ring) {
        case 8: scoped<8>(); break;
        case 10: scoped<10>(); break;
        case 16: scoped<16>(); break;
        default: scoped<12>(); break;
    }
}
void entry_replay() {
    const Launch& a = g_launch;
    k_trace_replay_persistent<12, false, false>(a.sc, a.rays, a.start, nullptr, nullptr, a.n, a.out, a.spill, a.total_threads, a.ticket,
                                                a.ticket_next, a.refill, a.ticket_size);
}
void entry_replay_queued() {          // the redo pass of EXACT: ray indices from the list, its length from the counter
    const Launch& a = g_launch;
    k_trace_replay_persistent<12, true, false>(a.sc, a.rays, a.start, a.queue, a.count_ptr, a.n, a.out, a.spill, a.total_threads, a.ticket,
                                               a.ticket_next, a.refill, a.ticket_size);
}
void entry_scoped_queued() {          // the integrator's form: ray positions from a queue, its length from a device word, start triangle in pad0
    const Launch& a = g_launch;
    k_trace_scoped_persistent<12, true>(a.sc, a.rays, nullptr, a.queue, a.count_ptr, a.n, a.out, a.spill, a.total_threads, a.ticket,
                                        a.ticket_next, a.refill, a.ticket_size, a.redo_list, a.redo_count, a.redo_count_next);
}
void entry_replay_redo_of_queued() {
    const Launch& a = g_launch;
    k_trace_replay_persistent<12, true, true>(a.sc, a.rays, nullptr, a.queue, a.count_ptr, a.n, a.out, a.spill, a.total_threads, a.ticket,
                                              a.ticket_next, a.refill, a.ticket_size);
}
void entry_mint() {
    const Launch& a = g_launch;
    k_trace_mint_persistent<12, false>(a.sc, a.rays, a.start, nullptr, nullptr, a.n, a.out, a.spill, a.total_threads, a.ticket,
                                       a.ticket_next, a.refill, a.ticket_size);
}

}  // namespace

extern "C" {

// One pt_trace_closest_dev call (or, with `queue`, one traversal launch of the integrator) as trace.hip's launcher makes it, on a grid of `blocks` blocks (a multiple of PTC_SHARDS), waves
// scheduled from `seed`.  mode: PT_TRAVERSE_MIN_T / PT_TRAVERSE_EXACT (the scoped kernel, then the frame replay over its redo
// list on a quarter of the grid) / PT_TRAVERSE_REPLAY.  out[i]: the hits.  stats: [0] fiber switches, [1] rays the scoped kernel
// listed for the redo pass.  Returns 0, or -1 with a message on stderr (lanes of a wave disagreeing on a collective, a block
// that does not end).
int pte_trace(uint32_t mode, const pt_bvh_node* nodes, uint32_t n_nodes, const pt_triangle* tris, uint32_t n_tris,
              const pt_ray* rays, const uint32_t* start, uint32_t n, pt_hit* out, uint32_t blocks, uint32_t refill,
              uint32_t ticket_size, uint64_t seed, uint64_t stats[2], const uint32_t* queue, uint32_t n_queue, uint32_t ring) {
    g_ring = ring;
    // queue != NULL (EXACT only): the kernels trace rays[queue[0 .. n_queue)] (n_queue is handed over through a device word, the
    // launch's n is the capacity), the start triangle rides in pad0 and hits are written by position; the rest of out[] keeps
    // the 0xDEADBEEF it was filled with
    if (queue && mode != PT_TRAVERSE_EXACT) { std::fprintf(stderr, "emu: a queue goes with PT_TRAVERSE_EXACT\n"); return -1; }
    if (blocks == 0 || blocks % PTC_SHARDS != 0) { std::fprintf(stderr, "emu: blocks must be a multiple of %u\n", PTC_SHARDS); return -1; }
    std::vector<float4> wide((size_t)(n_nodes ? n_nodes : 1) * 8, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    std::vector<float4> bin((size_t)(n_nodes ? n_nodes : 1) * 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t i = 0; i < n_nodes; ++i) if (!nodes[i].is_leaf) { make_wide(nodes, i, &wide[(size_t)i * 8]); make_node(nodes, i, &bin[(size_t)i * 4]); }
    std::vector<float4> trec((size_t)(n_tris ? n_tris : 1) * PTC_TRI_STRIDE);
    for (uint32_t i = 0; i < n_tris; ++i) precompute(tris[i], &trec[(size_t)i * PTC_TRI_STRIDE]);
    DevSceneView v{};
    v.nodes = bin.data();
    v.wide = wide.data();
    v.tris = trec.data();
    v.n_tris = n_nodes ? n_tris : 0u;
    v.root_ref = n_nodes && nodes[0].is_leaf ? (PTC_LEAF_BIT | nodes[0].shape_index) : 0u;
    v.max_depth = depth_of(nodes, n_nodes);
    v.wide_depth = v.max_depth / 2u + 1u;
    v.sign_select_ok = 1u;
    v.joins_exact = 1u;
    const uint32_t total_threads = blocks * PTC_BLOCK;
    // exact sizes, as the launcher reserves them, so that the sanitizer sees an access one past any of them
    std::vector<uint2> spill(ptc_spill_depth(v, mode) * total_threads);
    std::vector<uint32_t> tickets(PTC_TICKET_BYTES / 4u, 0u), redo_list(n ? n : 1u);
    std::vector<uint4> hits(n ? n : 1u);
    for (auto& h : hits) h = make_uint4(0xDEADBEEFu, 0u, 0u, 0u);
    g_launch = Launch{v, reinterpret_cast<const float4*>(rays), start, nullptr, nullptr, n, hits.data(), spill.data(), total_threads,
                      tickets.data(), tickets.data() + 512u, refill, ticket_size, redo_list.data(), tickets.data() + 1024u,
                      tickets.data() + 1024u + 64u};
    uint32_t queue_count = n_queue;
    if (queue) { g_launch.queue = queue; g_launch.count_ptr = &queue_count; }
    std::mt19937_64 rng(seed);
    emu::g_switches = 0;
    auto run_grid = [&](void (*entry)(), uint32_t grid) {
        for (uint32_t b = 0; b < grid; ++b) {
            if (!emu::run_block(entry, b, grid, rng, 1ull << 26)) {
                std::fprintf(stderr, "emu: block %u: %s\n", b, emu::g_error ? emu::g_error : "?");
                return false;
            }
        }
        return true;
    };
    stats[1] = 0;
    if (n != 0) {
        if (!run_grid(queue ? entry_scoped_queued : mode == PT_TRAVERSE_REPLAY ? entry_replay : mode == PT_TRAVERSE_MIN_T ? entry_mint : entry_scoped, blocks)) return -1;
        if (mode == PT_TRAVERSE_EXACT) {
            stats[1] = tickets[1024u];
            if (stats[1] > n) { std::fprintf(stderr, "emu: %llu rays on a redo list of %u\n", (unsigned long long)stats[1], n); return -1; }
            // the redo pass: the other set of ticket words (this launch cleared them), a quarter of the grid
            g_launch.ticket = tickets.data() + 512u;
            g_launch.ticket_next = tickets.data();
            g_launch.queue = redo_list.data();
            g_launch.count_ptr = tickets.data() + 1024u;
            uint32_t grid = blocks / 4u / PTC_SHARDS * PTC_SHARDS;
            if (grid < PTC_SHARDS) grid = PTC_SHARDS;
            if (!run_grid(queue ? entry_replay_redo_of_queued : entry_replay_queued, grid)) return -1;
        }
    }
    std::memcpy(out, hits.data(), (size_t)n * sizeof(pt_hit));
    stats[0] = emu::g_switches;
    return 0;
}

}  // extern "C"
