// host_bvh.cpp — pt_bvh_build: the host-side BVH builder a non-Rust host needs.
//
// In the target product the Rust host keeps BVH::build (bvh.rs:346-356 -> BVHNode::build
// bvh.rs:134-248) and hands `bvh.nodes` to pt_scene_create.  This image has no Rust, so the C++
// harness needs the same builder; it must reproduce the reference's split decisions bit for bit
// because topology decides traversal order, tie-breaks and visit counts.
//
// Same decisions, different construction: the reference recurses and allocates six Vec<usize> per
// node; here one index array is partitioned in place with a stable counting sort over the six
// buckets, triangle boxes/centroids are computed once, and node indices are known up front — in a
// preorder vector a subtree with k leaves occupies 2k-1 consecutive entries, so a node at index i
// with kL leaves on the left has children i+1 and i+2*kL — which lets independent subtrees be
// built by worker threads in any order.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ptcore.h"
#include "guard.h"
#include "../../include/pt_math.h"

namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr float kSplitEps = 0.00001f;          // bvh.rs:28
constexpr int kBuckets = 6;                    // bvh.rs:186

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = kInf; hi[k] = -kInf; } }          // bvh.rs:47-52
    void add(const Box& o) {                                                                // bvh.rs:54-67
        for (int k = 0; k < 3; ++k) { lo[k] = pt_fmin(lo[k], o.lo[k]); hi[k] = pt_fmax(hi[k], o.hi[k]); }
    }
    void add_point(const float p[3]) {                                                      // bvh.rs:82-87
        for (int k = 0; k < 3; ++k) { lo[k] = pt_fmin(lo[k], p[k]); hi[k] = pt_fmax(hi[k], p[k]); }
    }
    bool empty() const { return lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]; }          // bvh.rs:97-99
    float area() const {                                                                    // bvh.rs:101-104
        float sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
        return 2.0f * (sx * sy + sx * sz + sy * sz);
    }
};

struct Prim { Box box; float centre[3]; };

struct Job { uint32_t lo, hi, node; };

struct Builder {
    const std::vector<Prim>& prims;
    std::vector<uint32_t>& order;      // triangle indices, partitioned in place
    std::vector<uint32_t>& scratch;    // same size, counting-sort target
    std::vector<uint8_t>& bucket_of;   // per position in `order`
    pt_bvh_node* nodes;
    std::atomic<int> error{PT_OK};

    // work queue shared by the worker threads
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Job> queue;
    int busy = 0;
    bool done = false;
    uint32_t spawn_threshold = 1u << 14;

    Builder(const std::vector<Prim>& p, std::vector<uint32_t>& o, std::vector<uint32_t>& s,
            std::vector<uint8_t>& b, pt_bvh_node* n)
        : prims(p), order(o), scratch(s), bucket_of(b), nodes(n) {}

    static void store_box(pt_aabb& dst, const Box& b) {
        for (int k = 0; k < 3; ++k) { dst.min[k] = b.lo[k]; dst.max[k] = b.hi[k]; }
    }

    // Splits order[lo, hi) as BVHNode::build does and writes nodes[node]; returns the number of
    // leaves that went left (0 for a leaf node).
    uint32_t split(uint32_t lo, uint32_t hi, uint32_t node) {
        const uint32_t count = hi - lo;
        pt_bvh_node& out = nodes[node];
        std::memset(&out, 0, sizeof(out));
        if (count == 1) {                                                                   // bvh.rs:154-159
            out.is_leaf = 1;
            out.shape_index = order[lo];
            return 0;
        }
        Box hull, chull;                                                                    // bvh.rs:147-151
        hull.reset(); chull.reset();
        for (uint32_t i = lo; i < hi; ++i) {
            const Prim& p = prims[order[i]];
            hull.add(p.box);
            chull.add_point(p.centre);
        }
        // largest_axis(): first strict maximum of the centroid extent (bvh.rs:106-108,167)
        float ext[3] = {chull.hi[0] - chull.lo[0], chull.hi[1] - chull.lo[1], chull.hi[2] - chull.lo[2]};
        int axis = 0;
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        const float extent = chull.hi[axis] - chull.lo[axis];                               // bvh.rs:168

        Box lbox, rbox;
        uint32_t n_left;
        if (extent < kSplitEps) {                                                           // bvh.rs:171-183
            n_left = count / 2;
            lbox.reset(); rbox.reset();
            for (uint32_t i = lo; i < lo + n_left; ++i) lbox.add(prims[order[i]].box);
            for (uint32_t i = lo + n_left; i < hi; ++i) rbox.add(prims[order[i]].box);
        } else {
            Box bbox[kBuckets];
            uint32_t bcount[kBuckets] = {0, 0, 0, 0, 0, 0};
            for (auto& b : bbox) b.reset();
            const float origin = chull.lo[axis];
            const float scale = (float)kBuckets - 0.01f;                                    // bvh.rs:202
            for (uint32_t i = lo; i < hi; ++i) {                                            // bvh.rs:192-207
                const Prim& p = prims[order[i]];
                float rel = (p.centre[axis] - origin) / extent;
                uint32_t b = pt_f2usize(rel * scale);
                if (b >= (uint32_t)kBuckets) { error.store(PT_ERR_INVALID); b = kBuckets - 1; }
                bucket_of[i] = (uint8_t)b;
                bcount[b] += 1;
                bbox[b].add(p.box);
            }
            // cheapest of the five prefix/suffix splits; strict '<' keeps the first minimum
            int best = 0;                                                                   // bvh.rs:210-228
            float best_cost = kInf;
            lbox.reset(); rbox.reset();
            const float parent_area = hull.area();
            for (int s = 0; s < kBuckets - 1; ++s) {
                Box l, r; l.reset(); r.reset();
                uint32_t nl = 0, nr = 0;
                for (int k = 0; k <= s; ++k) { l.add(bbox[k]); nl += bcount[k]; }
                for (int k = s + 1; k < kBuckets; ++k) { r.add(bbox[k]); nr += bcount[k]; }
                float cost = ((float)nl * l.area() + (float)nr * r.area()) / parent_area;
                if (cost < best_cost) { best = s; best_cost = cost; lbox = l; rbox = r; }
            }
            if (lbox.empty() || rbox.empty()) {                                             // bvh.rs:242-243
                error.store(PT_ERR_BVH_DEGENERATE);
                // keep the node array well formed so callers never chase garbage
                out.is_leaf = 1; out.shape_index = order[lo];
                return 0;
            }
            // children = buckets concatenated in bucket order, original order kept inside each
            // bucket (bvh.rs:231-233): a stable counting sort of the range by bucket number
            uint32_t start[kBuckets];
            uint32_t acc = lo;
            for (int k = 0; k < kBuckets; ++k) { start[k] = acc; acc += bcount[k]; }
            for (uint32_t i = lo; i < hi; ++i) scratch[start[bucket_of[i]]++] = order[i];
            std::memcpy(&order[lo], &scratch[lo], sizeof(uint32_t) * count);
            n_left = 0;
            for (int k = 0; k <= best; ++k) n_left += bcount[k];
        }
        out.is_leaf = 0;
        out.child_l_index = node + 1;
        out.child_r_index = node + 2 * n_left;
        store_box(out.child_l_aabb, lbox);
        store_box(out.child_r_aabb, rbox);
        return n_left;
    }

    // Depth-first over one subtree with an explicit stack; large right subtrees are handed to
    // the shared queue when `share` is set.
    void run_subtree(Job root, bool share) {
        std::vector<Job> stack;
        stack.push_back(root);
        while (!stack.empty()) {
            Job j = stack.back();
            stack.pop_back();
            uint32_t n_left = split(j.lo, j.hi, j.node);
            if (nodes[j.node].is_leaf) continue;
            Job left{j.lo, j.lo + n_left, j.node + 1};
            Job right{j.lo + n_left, j.hi, j.node + 2 * n_left};
            if (share && (right.hi - right.lo) >= spawn_threshold) {
                { std::lock_guard<std::mutex> g(mu); queue.push_back(right); }
                cv.notify_one();
            } else {
                stack.push_back(right);
            }
            stack.push_back(left);
        }
    }

    void worker() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return done || !queue.empty(); });
                if (queue.empty()) return;
                j = queue.back();
                queue.pop_back();
                busy += 1;
            }
            run_subtree(j, true);
            {
                std::lock_guard<std::mutex> g(mu);
                busy -= 1;
                if (busy == 0 && queue.empty()) { done = true; cv.notify_all(); }
            }
        }
    }
};

}  // namespace

static int pt_bvh_build_body(const pt_triangle* triangles, uint32_t n, pt_bvh_node* nodes_out,
                            uint32_t* n_nodes_out) {
    if (n_nodes_out) *n_nodes_out = 0;
    if (n == 0) return PT_OK;                                                               // bvh.rs:353-355
    if (!triangles || !nodes_out) return PT_ERR_INVALID;
    if (n >= 0x40000000u) return PT_ERR_UNSUPPORTED;

    std::vector<Prim> prims(n);
    for (uint32_t i = 0; i < n; ++i) {
        const pt_triangle& t = triangles[i];
        Prim& p = prims[i];
        for (int k = 0; k < 3; ++k) {
            if (!(pt_fabs(t.a[k]) < kInf) || !(pt_fabs(t.b[k]) < kInf) || !(pt_fabs(t.c[k]) < kInf))
                return PT_ERR_INVALID;             // NaN / infinite vertex
            p.box.lo[k] = t.a[k]; p.box.hi[k] = t.a[k];                                     // triangle.rs:118-122
        }
        p.box.add_point(t.b);
        p.box.add_point(t.c);
        for (int k = 0; k < 3; ++k)                                                         // bvh.rs:93-95
            p.centre[k] = p.box.lo[k] + ((p.box.hi[k] - p.box.lo[k]) / 2.0f);
    }
    std::vector<uint32_t> order(n), scratch(n);
    std::vector<uint8_t> bucket_of(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;

    Builder b(prims, order, scratch, bucket_of, nodes_out);
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::min<unsigned>(hw ? hw : 1u, 16u);
    if (n < (1u << 16) || nthreads <= 1) {
        b.run_subtree(Job{0, n, 0}, false);
    } else {
        b.queue.push_back(Job{0, n, 0});
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back([&b] { b.worker(); });
        for (auto& t : pool) t.join();
    }
    if (n_nodes_out) *n_nodes_out = 2 * n - 1;
    return b.error.load();
}

extern "C" int pt_bvh_build(const pt_triangle* triangles, uint32_t n, pt_bvh_node* nodes_out,
                            uint32_t* n_nodes_out) {
    return ptc_noexcept_code([&] { return pt_bvh_build_body(triangles, n, nodes_out, n_nodes_out); });
}
