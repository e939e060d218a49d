// lbvh.hip — device-side BVH build.  No counterpart in the reference, whose
// BVH::build (bvh.rs:134-248) is a host-side binned-SAH recursion; pt_bvh_build restates that one bit
// for bit and stays the default.  This builder is for callers who want the tree made on the GPU:
//
//   1. scene bounds of the triangle boxes (ordered-int atomics)
//   2. 63-bit Morton code of every box centre (21 bits per axis), radix sort of (code, triangle) pairs
//   3. Karras' binary radix tree over the sorted codes (ties broken by position)
//   4. boxes bottom-up, one launch per tree level: a node is written once both children were written by
//      an earlier launch (the boxes of BOTH children live in the parent, device_types.h)
//   5. depth of every internal node, then the 4-wide records (device_types.h): internal nodes at odd
//      depth (root = 1) become wide nodes, numbered by a prefix sum in node order (root stays 0)
//
// Leaf boxes are the exact min/max of the triangle's vertices, as Triangle::aabb (triangle.rs:118-122);
// the tree's TOPOLOGY differs from the reference builder's, so traversal order differs; the minimum-t
// hit is the same (every triangle is in exactly one leaf, every box bounds its subtree), only exact
// equal-t ties can resolve differently.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <vector>

#include "device_types.h"
#include "../../include/pt_math.h"

#define LB_BLOCK 256

namespace {

// order-preserving float <-> uint (for atomicMin/atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}

// bounds[0..2] = min of box centres (ordered), bounds[3..5] = max
__global__ __launch_bounds__(LB_BLOCK) void k_bounds(const float4* __restrict__ box, uint32_t n, uint32_t* __restrict__ bounds) {
    float lo[3] = {pt_inf(), pt_inf(), pt_inf()}, hi[3] = {-pt_inf(), -pt_inf(), -pt_inf()};
    for (uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x; i < n; i += gridDim.x * LB_BLOCK) {
        const float4 a = box[2u * i], b = box[2u * i + 1u];
        const float c[3] = {0.5f * a.x + 0.5f * b.x, 0.5f * a.y + 0.5f * b.y, 0.5f * a.z + 0.5f * b.z};
        for (int k = 0; k < 3; ++k) {
            if (c[k] < lo[k]) lo[k] = c[k];
            if (c[k] > hi[k]) hi[k] = c[k];
        }
    }
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        if ((threadIdx.x & 63u) == 0u) {
            atomicMin(&bounds[k], f2ord(lo[k]));
            atomicMax(&bounds[3 + k], f2ord(hi[k]));
        }
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {       // 21 bits -> every third bit
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(LB_BLOCK) void k_morton(const float4* __restrict__ box, uint32_t n, const uint32_t* __restrict__ bounds,
                                                     uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 a = box[2u * i], b = box[2u * i + 1u];
    const float c[3] = {0.5f * a.x + 0.5f * b.x, 0.5f * a.y + 0.5f * b.y, 0.5f * a.z + 0.5f * b.z};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = ord2f(bounds[k]), hi = ord2f(bounds[3 + k]);
        const float ext = hi - lo;
        float t = ext > 0.0f ? (c[k] - lo) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);             // NaN centres (degenerate input) land in cell 0
        q[k] = (uint32_t)(t * 2097151.0f);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}

// Karras 2012: common-prefix length of sorted keys i and j; equal keys fall back to the positions
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, uint32_t n, int i, int j) {
    if (j < 0 || j >= (int)n) return -1;
    const uint64_t x = keys[i] ^ keys[j];
    if (x) return __clzll((long long)x);
    return 64 + __clz((uint32_t)i ^ (uint32_t)j);
}

// one thread per internal node i in [0, n-1): children + parent links.
// child refs use the device encoding: PTC_LEAF_BIT | triangle, or the internal node's index
__global__ __launch_bounds__(LB_BLOCK) void k_radix_tree(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                         uint2* __restrict__ children, uint32_t* __restrict__ parent_of_internal,
                                                         uint32_t* __restrict__ parent_of_leaf) {
    const int i = (int)(blockIdx.x * LB_BLOCK + threadIdx.x);
    if (i >= (int)n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int div = 2, t = (l + 1) / 2;; div *= 2, t = (l + div - 1) / div) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int split = i + s * d + (d < 0 ? -1 : 0);
    const int first = d > 0 ? i : j, last = d > 0 ? j : i;
    uint32_t lref, rref;
    if (split == first) { lref = PTC_LEAF_BIT | vals[split]; parent_of_leaf[split] = (uint32_t)i; }
    else { lref = (uint32_t)split; parent_of_internal[split] = (uint32_t)i; }
    if (split + 1 == last) { rref = PTC_LEAF_BIT | vals[split + 1]; parent_of_leaf[split + 1] = (uint32_t)i; }
    else { rref = (uint32_t)(split + 1); parent_of_internal[split + 1] = (uint32_t)i; }
    children[i] = make_uint2(lref, rref);
    if (i == 0) parent_of_internal[0] = 0xFFFFFFFFu;
}

// Boxes bottom-up, one LAUNCH per tree level instead of one kernel with arrival counters: a node is written in
// round r when both children were written in an EARLIER launch (stamp[c] in 1..r), so everything a thread reads
// was published by a kernel boundary.  (The single-kernel form needs a device-scope fence per step, which on
// this part writes back the L2: that kernel alone took 162 ms for 20 M triangles; the whole build now takes 48 ms.)
// stamp[i] = round + 1 once node i is written; left[round] counts the nodes still waiting after that round.
__global__ __launch_bounds__(LB_BLOCK) void k_fit_round(const float4* __restrict__ box, uint32_t n_internal, uint32_t round,
                                                        const uint2* __restrict__ children, uint32_t* __restrict__ stamp,
                                                        float4* __restrict__ node_box, float4* __restrict__ nodes,
                                                        uint32_t* __restrict__ left) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    bool waiting = false;
    if (i < n_internal && stamp[i] == 0u) {
        const uint2 ch = children[i];
        auto ready = [&](uint32_t ref) {
            if (ref & PTC_LEAF_BIT) return true;
            const uint32_t st = stamp[ref];
            return st != 0u && st <= round;
        };
        if (ready(ch.x) && ready(ch.y)) {
            float4 b[4];
            for (int side = 0; side < 2; ++side) {
                const uint32_t ref = side ? ch.y : ch.x;
                const float4* src = (ref & PTC_LEAF_BIT) ? &box[2u * (ref & PTC_INDEX_MASK)] : &node_box[2u * ref];
                b[2 * side] = src[0];
                b[2 * side + 1] = src[1];
            }
            float4* q = &nodes[4u * (size_t)i];
            q[0] = make_float4(b[0].x, b[0].y, b[0].z, __uint_as_float(ch.x));
            q[1] = make_float4(b[1].x, b[1].y, b[1].z, __uint_as_float(ch.y));
            q[2] = make_float4(b[2].x, b[2].y, b[2].z, 0.0f);
            q[3] = make_float4(b[3].x, b[3].y, b[3].z, 0.0f);
            // AABB::join (bvh.rs:50-62): component-wise min / max
            node_box[2u * i] = make_float4(fminf(b[0].x, b[2].x), fminf(b[0].y, b[2].y), fminf(b[0].z, b[2].z), 0.0f);
            node_box[2u * i + 1u] = make_float4(fmaxf(b[1].x, b[3].x), fmaxf(b[1].y, b[3].y), fmaxf(b[1].z, b[3].z), 0.0f);
            stamp[i] = round + 1u;
        } else waiting = true;
    }
    const unsigned long long m = __ballot(waiting);
    if ((threadIdx.x & 63u) == 0u && m) atomicAdd(&left[round], (uint32_t)__popcll(m));
}

// depth of internal node i (root = 1); is_wide[i] = depth odd; stats[0] = max depth
__global__ __launch_bounds__(LB_BLOCK) void k_depth(const uint32_t* __restrict__ parent_of_internal, uint32_t n_internal,
                                                    uint32_t* __restrict__ is_wide, uint32_t* __restrict__ stats) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n_internal) return;
    uint32_t d = 1u;            // (the walk gives up past PTC_MAX_BVH_DEPTH: the caller rejects such a tree)
    for (uint32_t c = parent_of_internal[i]; c != 0xFFFFFFFFu && d <= PTC_MAX_BVH_DEPTH; c = parent_of_internal[c]) ++d;
    is_wide[i] = d & 1u;
    atomicMax(&stats[0], d);
}

// wide record of every odd-depth internal node (layout: device_types.h)
__global__ __launch_bounds__(LB_BLOCK) void k_wide(const float4* __restrict__ nodes, const uint32_t* __restrict__ is_wide,
                                                   const uint32_t* __restrict__ wide_index, uint32_t n_internal, float4* __restrict__ wide) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n_internal || !is_wide[i]) return;
    const float nan = __uint_as_float(0x7FC00000u);
    float lo[3][4], hi[3][4];
    uint32_t refs[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 4; ++k) { lo[a][k] = nan; hi[a][k] = nan; }
    auto take = [&](const float4* q, int side, int slot) {
        const float4 l = q[side ? 2 : 0], h = q[side ? 3 : 1];
        lo[0][slot] = l.x; lo[1][slot] = l.y; lo[2][slot] = l.z; hi[0][slot] = h.x; hi[1][slot] = h.y; hi[2][slot] = h.z;
        const uint32_t r = __float_as_uint(side ? q[1].w : q[0].w);
        refs[slot] = (r & PTC_LEAF_BIT) ? r : wide_index[r];
    };
    // slots at fixed places in REVERSE binary order (RR, RL | LR, LL), a leaf child in the first slot of its pair, as the
    // host-side collapse stores them (core.hip, device_types.h)
    const float4* q = &nodes[4u * (size_t)i];
#pragma unroll
    for (int side = 1; side >= 0; --side) {
        const int base = side ? 0 : 2;
        const uint32_t r = __float_as_uint(side ? q[1].w : q[0].w);
        if (r & PTC_LEAF_BIT) take(q, side, base);
        else { const float4* q2 = &nodes[4u * (size_t)r]; take(q2, 1, base); take(q2, 0, base + 1); }
    }
    float4* w = &wide[8u * (size_t)wide_index[i]];
    for (int a = 0; a < 3; ++a) {
        w[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
        w[3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
    }
    w[6] = make_float4(__uint_as_float(refs[0]), __uint_as_float(refs[1]), __uint_as_float(refs[2]), __uint_as_float(refs[3]));
    w[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// The EXACT traversal does not read the boxes of an odd-depth node's two children: it re-derives each as the join of that
// child's own two child boxes (trace_step.h).  True of every tree BVH::build or the LBVH makes; a caller's tree is checked
// here: stats[1] = 1 if any such stored box differs from the minNum / maxNum join (the sign of a zero aside; NaN == NaN).
__global__ __launch_bounds__(LB_BLOCK) void k_check_joins(const float4* __restrict__ nodes, const uint32_t* __restrict__ is_wide,
                                                          uint32_t n_internal, uint32_t* __restrict__ stats) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n_internal || !is_wide[i]) return;
    const float4* q = &nodes[4u * (size_t)i];
    bool bad = false;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const uint32_t r = __float_as_uint(side ? q[1].w : q[0].w);
        if (r & PTC_LEAF_BIT) continue;
        const float4 lo = q[side ? 2 : 0], hi = q[side ? 3 : 1];
        const float4* c = &nodes[4u * (size_t)r];
        const float jl[3] = {fminf(c[0].x, c[2].x), fminf(c[0].y, c[2].y), fminf(c[0].z, c[2].z)};
        const float jh[3] = {fmaxf(c[1].x, c[3].x), fmaxf(c[1].y, c[3].y), fmaxf(c[1].z, c[3].z)};
        const float sl[3] = {lo.x, lo.y, lo.z}, sh[3] = {hi.x, hi.y, hi.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!(sl[k] == jl[k] || (sl[k] != sl[k] && jl[k] != jl[k]))) bad = true;
            if (!(sh[k] == jh[k] || (sh[k] != sh[k] && jh[k] != jh[k]))) bad = true;
        }
    }
    if (bad) atomicOr(&stats[1], 1u);
}

// parent links of a finished binary node array (caller-supplied trees: the links are not known yet)
__global__ __launch_bounds__(LB_BLOCK) void k_parents(const float4* __restrict__ nodes, uint32_t n_internal,
                                                      uint32_t* __restrict__ parent_of_internal) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n_internal) return;
    const uint32_t l = __float_as_uint(nodes[4u * (size_t)i].w), r = __float_as_uint(nodes[4u * (size_t)i + 1u].w);
    if (!(l & PTC_LEAF_BIT)) parent_of_internal[l] = i;
    if (!(r & PTC_LEAF_BIT)) parent_of_internal[r] = i;
    if (i == 0u) parent_of_internal[0] = 0xFFFFFFFFu;
}

// ---- caller-supplied node vectors (pt_scene_create): validate and flatten on the device -------------------------------
// err bits: 1 leaf shape_index out of range, 2 child index not a forward preorder index, 4 node referenced by two parents
// (so another is unreachable), 8 triangle referenced by two leaves, 16 (not an error) a child box with min > max or with a
// NaN in only one of min / max on an axis
__global__ __launch_bounds__(LB_BLOCK) void k_node_flags(const uint4* __restrict__ raw, uint32_t n_nodes, uint32_t n_tris,
                                                         uint32_t* __restrict__ is_internal, uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n_nodes) return;
    const uint4 h = raw[4u * (size_t)i];            // {is_leaf, shape_index, child_l_index, child_r_index}
    uint32_t bad = 0u;
    if (h.x) { if (h.y >= n_tris) bad = 1u; }
    else if (h.z <= i || h.z >= n_nodes || h.w <= i || h.w >= n_nodes) bad = 2u;
    is_internal[i] = h.x ? 0u : 1u;
    if (bad) atomicOr(err, bad);
}

// one {child boxes, child refs} record per internal node, numbered by rank among internal nodes (device_types.h); every
// child marks its bit: with forward indices, 2N - 2 references and no bit marked twice, every node but the root has exactly
// one parent and every triangle exactly one leaf
__global__ __launch_bounds__(LB_BLOCK) void k_node_pack(const uint4* __restrict__ raw, uint32_t n_nodes, const uint32_t* __restrict__ is_internal,
                                                        const uint32_t* __restrict__ rank, uint32_t* __restrict__ node_bits,
                                                        uint32_t* __restrict__ tri_bits, float4* __restrict__ nodes, uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n_nodes || !is_internal[i]) return;
    const uint4 h = raw[4u * (size_t)i], b0 = raw[4u * (size_t)i + 1u], b1 = raw[4u * (size_t)i + 2u], b2 = raw[4u * (size_t)i + 3u];
    uint32_t bad = 0u, refs[2];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const uint32_t c = side ? h.w : h.z;
        const uint4 ch = raw[4u * (size_t)c];
        if (atomicOr(&node_bits[c >> 5], 1u << (c & 31u)) & (1u << (c & 31u))) bad |= 4u;
        if (ch.x) {
            refs[side] = PTC_LEAF_BIT | ch.y;
            if (atomicOr(&tri_bits[ch.y >> 5], 1u << (ch.y & 31u)) & (1u << (ch.y & 31u))) bad |= 8u;
        } else refs[side] = rank[c];
    }
    // child_l_aabb = {b0.x b0.y b0.z | b0.w b1.x b1.y}, child_r_aabb = {b1.z b1.w b2.x | b2.y b2.z b2.w}
    const float lmin[3] = {__uint_as_float(b0.x), __uint_as_float(b0.y), __uint_as_float(b0.z)}, lmax[3] = {__uint_as_float(b0.w), __uint_as_float(b1.x), __uint_as_float(b1.y)};
    const float rmin[3] = {__uint_as_float(b1.z), __uint_as_float(b1.w), __uint_as_float(b2.x)}, rmax[3] = {__uint_as_float(b2.y), __uint_as_float(b2.z), __uint_as_float(b2.w)};
#pragma unroll
    for (int k = 0; k < 3; ++k)         // ordered: min <= max, or both NaN (a NaN on one side only breaks the sign-select step too)
        if (!(lmin[k] <= lmax[k] || (lmin[k] != lmin[k] && lmax[k] != lmax[k])) || !(rmin[k] <= rmax[k] || (rmin[k] != rmin[k] && rmax[k] != rmax[k]))) bad |= 16u;
    float4* q = &nodes[4u * (size_t)rank[i]];
    q[0] = make_float4(lmin[0], lmin[1], lmin[2], __uint_as_float(refs[0]));
    q[1] = make_float4(lmax[0], lmax[1], lmax[2], __uint_as_float(refs[1]));
    q[2] = make_float4(rmin[0], rmin[1], rmin[2], 0.0f);
    q[3] = make_float4(rmax[0], rmax[1], rmax[2], 0.0f);
    if (bad) atomicOr(err, bad);
}

struct Scratch {                     // frees what it holds when the build leaves, however it leaves
    std::vector<void*> p;
    ~Scratch() { for (void* q : p) (void)hipFree(q); }
    template <class T>
    int get(pt_ctx* ctx, T** out, size_t count, const char* what) {
        void* q = nullptr;
        const int rc = ptc_hip(ctx, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)), what);
        if (rc != PT_OK) return rc;
        p.push_back(q);
        *out = (T*)q;
        return PT_OK;
    }
};

}  // namespace

// boxes: 2 float4 per triangle {min.xyz,0},{max.xyz,0} (host).  On success *d_nodes ([n-1][4]) and *d_wide
// ([*n_wide][8]) are device arrays the caller owns (hipFree), depths as DevSceneView wants them.  n >= 2.
int ptc_build_lbvh(pt_ctx* ctx, const float4* h_boxes, uint32_t n, float4** d_nodes, float4** d_wide, uint32_t* n_wide,
                   uint32_t* max_depth, uint32_t* wide_depth, float* build_ms) {
    *d_nodes = nullptr; *d_wide = nullptr;
    hipStream_t s = ctx->stream;
    const uint32_t ni = n - 1u;
    Scratch tmp;
    float4 *box = nullptr, *node_box = nullptr;
    uint64_t *keys = nullptr, *keys2 = nullptr;
    uint32_t *vals = nullptr, *vals2 = nullptr, *bounds = nullptr, *par_i = nullptr, *par_l = nullptr, *stamp = nullptr, *is_wide = nullptr,
             *wide_index = nullptr, *stats = nullptr, *left = nullptr;
    uint2* children = nullptr;
    int rc;
    if ((rc = tmp.get(ctx, &box, 2 * (size_t)n, "lbvh boxes")) || (rc = tmp.get(ctx, &node_box, 2 * (size_t)ni, "lbvh node boxes")) ||
        (rc = tmp.get(ctx, &keys, n, "lbvh keys")) || (rc = tmp.get(ctx, &keys2, n, "lbvh keys")) ||
        (rc = tmp.get(ctx, &vals, n, "lbvh values")) || (rc = tmp.get(ctx, &vals2, n, "lbvh values")) ||
        (rc = tmp.get(ctx, &bounds, 8, "lbvh bounds")) || (rc = tmp.get(ctx, &par_i, ni, "lbvh parents")) ||
        (rc = tmp.get(ctx, &par_l, n, "lbvh parents")) || (rc = tmp.get(ctx, &stamp, ni, "lbvh stamps")) ||
        (rc = tmp.get(ctx, &is_wide, ni, "lbvh flags")) || (rc = tmp.get(ctx, &wide_index, ni, "lbvh wide index")) ||
        (rc = tmp.get(ctx, &stats, 4, "lbvh stats")) || (rc = tmp.get(ctx, &left, 512, "lbvh counters")) || (rc = tmp.get(ctx, &children, ni, "lbvh children")))
        return rc;
    float4* nodes = nullptr;
    PTC_HIP(ctx, hipMalloc((void**)&nodes, (size_t)ni * 4u * sizeof(float4)));
    struct Owned { float4* p; ~Owned() { if (p) (void)hipFree(p); } } own_nodes{nodes}, own_wide{nullptr};

    PTC_HIP(ctx, hipMemcpyAsync(box, h_boxes, 2 * (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s));
    PTC_HIP(ctx, hipEventRecord(ctx->ev[0], s));
    const uint32_t init_bounds[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    PTC_HIP(ctx, hipMemcpyAsync(bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, s));
    PTC_HIP(ctx, hipMemsetAsync(stamp, 0, (size_t)ni * sizeof(uint32_t), s));
    PTC_HIP(ctx, hipMemsetAsync(stats, 0, 4 * sizeof(uint32_t), s));
    PTC_HIP(ctx, hipMemsetAsync(left, 0, 512 * sizeof(uint32_t), s));
    const uint32_t gn = (n + LB_BLOCK - 1) / LB_BLOCK, gi = (ni + LB_BLOCK - 1) / LB_BLOCK;
    hipLaunchKernelGGL(k_bounds, dim3(std::min<uint32_t>(gn, (uint32_t)ctx->n_cus * 8u)), dim3(LB_BLOCK), 0, s, box, n, bounds);
    hipLaunchKernelGGL(k_morton, dim3(gn), dim3(LB_BLOCK), 0, s, box, n, bounds, keys, vals);
    size_t sort_bytes = 0;
    PTC_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys, keys2, vals, vals2, (int)n, 0, 63, s));
    void* sort_tmp = nullptr;
    if ((rc = tmp.get(ctx, (char**)&sort_tmp, sort_bytes, "lbvh sort workspace"))) return rc;
    PTC_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, keys, keys2, vals, vals2, (int)n, 0, 63, s));
    hipLaunchKernelGGL(k_radix_tree, dim3(gi), dim3(LB_BLOCK), 0, s, keys2, vals2, n, children, par_i, par_l);
    // one launch per level; the host looks at the waiting count every 8 rounds
    {
        constexpr uint32_t MAX_ROUNDS = 512, CHUNK = 8;
        bool done = false;
        for (uint32_t r0 = 0; r0 < MAX_ROUNDS && !done; r0 += CHUNK) {
            for (uint32_t r = r0; r < r0 + CHUNK; ++r)
                hipLaunchKernelGGL(k_fit_round, dim3(gi), dim3(LB_BLOCK), 0, s, box, ni, r, children, stamp, node_box, nodes, left);
            uint32_t h_left = 1;
            PTC_HIP(ctx, hipMemcpyAsync(&h_left, left + (r0 + CHUNK - 1u), 4, hipMemcpyDeviceToHost, s));
            PTC_HIP(ctx, hipStreamSynchronize(s));
            done = h_left == 0u;
        }
        if (!done) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "lbvh: tree deeper than 512 levels");
    }
    hipLaunchKernelGGL(k_depth, dim3(gi), dim3(LB_BLOCK), 0, s, par_i, ni, is_wide, stats);
    size_t scan_bytes = 0;
    PTC_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, is_wide, wide_index, (int)ni, s));
    void* scan_tmp = nullptr;
    if ((rc = tmp.get(ctx, (char**)&scan_tmp, scan_bytes, "lbvh scan workspace"))) return rc;
    PTC_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, is_wide, wide_index, (int)ni, s));
    // number of wide nodes = wide_index[last] + is_wide[last]; max depth from stats
    uint32_t h_last[2] = {0, 0}, h_stats[4] = {0, 0, 0, 0};
    PTC_HIP(ctx, hipMemcpyAsync(&h_last[0], wide_index + (ni - 1u), 4, hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipMemcpyAsync(&h_last[1], is_wide + (ni - 1u), 4, hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipMemcpyAsync(h_stats, stats, sizeof(h_stats), hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipStreamSynchronize(s));
    const uint32_t nw = h_last[0] + h_last[1];
    if (nw == 0 || h_stats[0] == 0) return ptc_fail(ctx, PT_ERR_HIP, "lbvh: build produced no nodes");
    float4* wide = nullptr;
    PTC_HIP(ctx, hipMalloc((void**)&wide, (size_t)nw * 8u * sizeof(float4)));
    own_wide.p = wide;
    hipLaunchKernelGGL(k_wide, dim3(gi), dim3(LB_BLOCK), 0, s, nodes, is_wide, wide_index, ni, wide);
    PTC_HIP(ctx, hipGetLastError());
    PTC_HIP(ctx, hipEventRecord(ctx->ev[1], s));
    PTC_HIP(ctx, hipStreamSynchronize(s));
    float ms = 0.0f;
    PTC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    if (build_ms) *build_ms = ms;
    *max_depth = h_stats[0];
    *wide_depth = (h_stats[0] + 1u) / 2u;
    *n_wide = nw;
    *d_nodes = nodes; *d_wide = wide;
    own_nodes.p = nullptr; own_wide.p = nullptr;
    return PT_OK;
}


// The 4-wide records of a binary node array that is already in HBM (the reference-topology tree flattened by
// pt_scene_create): depths, prefix sum over odd-depth nodes, k_wide — the same kernels the LBVH build ends with, and the
// same numbering (preorder among wide nodes) and child order as the host-side collapse in core.hip.
int ptc_build_wide_from_nodes(pt_ctx* ctx, const float4* d_nodes, uint32_t n_internal, float4** d_wide, uint32_t* n_wide,
                              uint32_t* max_depth, uint32_t* wide_depth, bool* joins_exact) {
    *d_wide = nullptr;
    *joins_exact = true;
    hipStream_t s = ctx->stream;
    Scratch tmp;
    uint32_t *par_i = nullptr, *is_wide = nullptr, *wide_index = nullptr, *stats = nullptr;
    int rc;
    if ((rc = tmp.get(ctx, &par_i, n_internal, "wide parents")) || (rc = tmp.get(ctx, &is_wide, n_internal, "wide flags")) ||
        (rc = tmp.get(ctx, &wide_index, n_internal, "wide index")) || (rc = tmp.get(ctx, &stats, 4, "wide stats")))
        return rc;
    PTC_HIP(ctx, hipMemsetAsync(stats, 0, 4 * sizeof(uint32_t), s));
    const uint32_t gi = (n_internal + LB_BLOCK - 1) / LB_BLOCK;
    hipLaunchKernelGGL(k_parents, dim3(gi), dim3(LB_BLOCK), 0, s, d_nodes, n_internal, par_i);
    hipLaunchKernelGGL(k_depth, dim3(gi), dim3(LB_BLOCK), 0, s, par_i, n_internal, is_wide, stats);
    hipLaunchKernelGGL(k_check_joins, dim3(gi), dim3(LB_BLOCK), 0, s, d_nodes, is_wide, n_internal, stats);
    size_t scan_bytes = 0;
    PTC_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, is_wide, wide_index, (int)n_internal, s));
    char* scan_tmp = nullptr;
    if ((rc = tmp.get(ctx, &scan_tmp, scan_bytes, "wide scan workspace"))) return rc;
    PTC_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, is_wide, wide_index, (int)n_internal, s));
    uint32_t h_last[2] = {0, 0}, h_stats[4] = {0, 0, 0, 0};
    PTC_HIP(ctx, hipMemcpyAsync(&h_last[0], wide_index + (n_internal - 1u), 4, hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipMemcpyAsync(&h_last[1], is_wide + (n_internal - 1u), 4, hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipMemcpyAsync(h_stats, stats, sizeof(h_stats), hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipStreamSynchronize(s));
    const uint32_t nw = h_last[0] + h_last[1];
    if (h_stats[0] > PTC_MAX_BVH_DEPTH) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_scene_create: BVH deeper than 4096 levels");
    if (nw == 0 || h_stats[0] == 0) return ptc_fail(ctx, PT_ERR_HIP, "wide collapse produced no nodes");
    float4* wide = nullptr;
    PTC_HIP(ctx, hipMalloc((void**)&wide, (size_t)nw * 8u * sizeof(float4)));
    hipLaunchKernelGGL(k_wide, dim3(gi), dim3(LB_BLOCK), 0, s, d_nodes, is_wide, wide_index, n_internal, wide);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(wide); return ptc_hip(ctx, e, "wide collapse"); }
    *d_wide = wide;
    *n_wide = nw;
    *max_depth = h_stats[0];
    *wide_depth = (h_stats[0] + 1u) / 2u;
    *joins_exact = h_stats[1] == 0u;
    return PT_OK;
}


// pt_scene_create's node vector, validated and flattened on the device: the raw 64-byte entries are uploaded as they are,
// the internal nodes are ranked by a prefix sum and packed into the binary node array of device_types.h.  Same verdicts as
// the host-side loop in core.hip (PTC_FLATTEN=1): every child index points forward, every node but the root has one
// parent, every triangle one leaf.  *d_nodes (n_internal x 4 float4) is the caller's on success.
int ptc_flatten_nodes(pt_ctx* ctx, const pt_bvh_node* h_nodes, uint32_t n_nodes, uint32_t n_tris, float4** d_nodes,
                      uint32_t* n_internal, bool* boxes_ordered) {
    static_assert(sizeof(pt_bvh_node) == 64, "k_node_flags / k_node_pack read a node as four 16-byte words");
    *d_nodes = nullptr; *n_internal = 0; *boxes_ordered = true;
    if (n_nodes == 0) return PT_OK;
    hipStream_t s = ctx->stream;
    Scratch tmp;
    uint4* raw = nullptr;
    uint32_t *is_internal = nullptr, *rank = nullptr, *node_bits = nullptr, *tri_bits = nullptr, *err = nullptr;
    int rc;
    const size_t nb_words = ((size_t)n_nodes + 31u) / 32u, tb_words = ((size_t)n_tris + 31u) / 32u;
    if ((rc = tmp.get(ctx, &raw, 4 * (size_t)n_nodes, "raw nodes")) || (rc = tmp.get(ctx, &is_internal, n_nodes, "node flags")) ||
        (rc = tmp.get(ctx, &rank, n_nodes, "node ranks")) || (rc = tmp.get(ctx, &node_bits, nb_words, "node marks")) ||
        (rc = tmp.get(ctx, &tri_bits, tb_words, "triangle marks")) || (rc = tmp.get(ctx, &err, 4, "node verdict")))
        return rc;
    PTC_HIP(ctx, hipMemcpyAsync(raw, h_nodes, (size_t)n_nodes * sizeof(pt_bvh_node), hipMemcpyHostToDevice, s));
    PTC_HIP(ctx, hipMemsetAsync(err, 0, 4 * sizeof(uint32_t), s));
    PTC_HIP(ctx, hipMemsetAsync(node_bits, 0, nb_words * sizeof(uint32_t), s));
    PTC_HIP(ctx, hipMemsetAsync(tri_bits, 0, std::max<size_t>(tb_words, 1) * sizeof(uint32_t), s));
    const uint32_t gn = (n_nodes + LB_BLOCK - 1) / LB_BLOCK;
    hipLaunchKernelGGL(k_node_flags, dim3(gn), dim3(LB_BLOCK), 0, s, raw, n_nodes, n_tris, is_internal, err);
    size_t scan_bytes = 0;
    PTC_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, is_internal, rank, (int)n_nodes, s));
    char* scan_tmp = nullptr;
    if ((rc = tmp.get(ctx, &scan_tmp, scan_bytes, "node scan workspace"))) return rc;
    PTC_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, is_internal, rank, (int)n_nodes, s));
    uint32_t h_last[2] = {0, 0}, h_err = 0;
    PTC_HIP(ctx, hipMemcpyAsync(&h_last[0], rank + (n_nodes - 1u), 4, hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipMemcpyAsync(&h_last[1], is_internal + (n_nodes - 1u), 4, hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipStreamSynchronize(s));
    if (h_err & 1u) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: leaf shape_index out of range");
    if (h_err & 2u) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: child index is not a forward preorder index");
    const uint32_t ni = h_last[0] + h_last[1];
    if (n_tris && ni != n_tris - 1u) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: expected n_triangles-1 internal nodes");
    *n_internal = ni;
    if (ni == 0) return PT_OK;                          // a single leaf
    float4* nodes = nullptr;
    PTC_HIP(ctx, hipMalloc((void**)&nodes, (size_t)ni * 4u * sizeof(float4)));
    hipLaunchKernelGGL(k_node_pack, dim3(gn), dim3(LB_BLOCK), 0, s, raw, n_nodes, is_internal, rank, node_bits, tri_bits, nodes, err);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(nodes); return ptc_hip(ctx, e, "node flattening"); }
    if (h_err & 12u) {
        (void)hipFree(nodes);
        return ptc_fail(ctx, PT_ERR_INVALID, (h_err & 4u) ? "pt_scene_create: node referenced by two parents (another is unreachable from the root)"
                                                         : "pt_scene_create: triangle referenced by two leaves");
    }
    *boxes_ordered = (h_err & 16u) == 0u;
    *d_nodes = nodes;
    return PT_OK;
}
