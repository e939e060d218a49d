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
// Blocks run one after the other (a valid schedule of a grid whose blocks only meet at the ticket words).  The launcher is
// restated here so that a test can choose grid, refill threshold, ticket size, ring size and schedule per launch; the library
// build (libptcore_emu.so, tests/test_emu_lib.py) runs trace.hip's own launcher.
#include "emu_rt.h"      // the fibers, the device vocabulary and the scheduler (shared with the library build, libptcore_emu.so)

#include <cmath>
#include <limits>

#include "../../pathtracer_amd/csrc/device_types.h"
#include "../../pathtracer_amd/csrc/trace_step.h"
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
    emu::g_rng.seed(seed);
    emu::g_switches = 0;
    emu::g_error = nullptr;
    auto run_grid = [&](void (*entry)(), uint32_t grid) {
        emu::g_entry = entry;
        for (uint32_t b = 0; b < grid; ++b) {
            if (!emu::run_block(b, grid, PTC_BLOCK, 1ull << 26)) {
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
