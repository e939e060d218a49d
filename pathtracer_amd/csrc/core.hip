// core.hip — context, scene upload (flattening to the device layouts) and the closest-hit entry
// points of the C ABI (include/ptcore.h).
//
// Scene upload replaces "hold Scene { bvh, triangles, materials, sky }" (scene.rs:185-191,
// 273-281): the caller's reference-order node vector (enum BVHNode, bvh.rs:123-127) becomes an
// internal-nodes-only array of 64-byte records, Triangle::new's precomputation
// (triangle.rs:50-60) is redone here in the reference's operation order, and materials / sky /
// spectral tables are copied to HBM once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "device_types.h"
#include "tables.h"
#include "../../include/pt_math.h"

static std::string g_init_error;
static std::mutex g_init_mu;

int ptc_fail(pt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    else { std::lock_guard<std::mutex> g(g_init_mu); g_init_error = msg; }
    return code;
}

int ptc_hip(pt_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return PT_OK;
    return ptc_fail(ctx, e == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP,
                    std::string(what) + ": " + hipGetErrorString(e));
}

int ptc_reserve(pt_ctx* ctx, void** buf, size_t* have, size_t need, const char* what) {
    if (*have >= need) return PT_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *have = 0; }
    size_t want = need + need / 4;
    hipError_t e = hipMalloc(buf, want);
    if (e != hipSuccess) { (void)hipGetLastError(); want = need; e = hipMalloc(buf, want); }
    if (e != hipSuccess) { (void)hipGetLastError(); *buf = nullptr; return ptc_hip(ctx, e, what); }   // (clears the sticky error: callers may retry smaller)
    *have = want;
    return PT_OK;
}

int ptc_timed_event(pt_ctx* ctx, size_t i) {
    while (ctx->trace_ev.size() <= i) {
        hipEvent_t e = nullptr;
        int rc = ptc_hip(ctx, hipEventCreate(&e), "hipEventCreate");
        if (rc != PT_OK) return rc;
        ctx->trace_ev.push_back(e);
    }
    return ptc_hip(ctx, hipEventRecord(ctx->trace_ev[i], ctx->stream), "hipEventRecord");
}

extern "C" int pt_abi_version(void) { return PT_ABI_VERSION; }

extern "C" const char* pt_last_error(const pt_ctx* ctx) {
    if (ctx) return ctx->error.c_str();
    std::lock_guard<std::mutex> g(g_init_mu);
    return g_init_error.c_str();
}

static int pt_init_body(const int* device_ids, int n, pt_ctx** out) {
    if (!out) return ptc_fail(nullptr, PT_ERR_INVALID, "pt_init: out is NULL");
    *out = nullptr;
    if (n != 1) return ptc_fail(nullptr, PT_ERR_INVALID, "pt_init: one context drives exactly one GPU (n must be 1)");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return ptc_fail(nullptr, PT_ERR_HIP, std::string("pt_init: no HIP device: ") + hipGetErrorString(e));
    int dev = device_ids ? device_ids[0] : 0;
    if (dev < 0 || dev >= count) return ptc_fail(nullptr, PT_ERR_INVALID, "pt_init: device id out of range");
    // a half-built context is torn down by pt_shutdown (every member starts out null)
    struct Guard { pt_ctx* p; ~Guard() { if (p) pt_shutdown(p); } } guard{new (std::nothrow) pt_ctx()};
    if (!guard.p) return ptc_fail(nullptr, PT_ERR_NOMEM, "pt_init: out of host memory");
    pt_ctx* ctx = guard.p;
    ctx->device = dev;
    auto bad = [&](const char* what) { return ptc_fail(nullptr, e == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP,
                                                       std::string("pt_init: ") + what + ": " + hipGetErrorString(e)); };
    if ((e = hipSetDevice(dev)) != hipSuccess) return bad("hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return bad("hipGetDeviceProperties");
    ctx->n_cus = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bad("hipStreamCreate");
    if ((e = hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking)) != hipSuccess) return bad("hipStreamCreate");
    if ((e = hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking)) != hipSuccess) return bad("hipStreamCreate");
    for (auto& ev : ctx->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bad("hipEventCreate");
    for (auto& ev : ctx->count_ev)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate");
    if ((e = hipHostMalloc((void**)&ctx->h_counters, 64 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) return bad("hipHostMalloc");
    if ((e = hipMalloc((void**)&ctx->ticket, PTC_TICKET_BYTES)) != hipSuccess) return bad("hipMalloc");
    if ((e = hipMemset(ctx->ticket, 0, PTC_TICKET_BYTES)) != hipSuccess) return bad("hipMemset");
    if (ptc_noalloc_setup(ctx) != PT_OK) return ptc_fail(nullptr, PT_ERR_HIP, "pt_init: " + ctx->error);
    auto env_u32 = [](const char* name, uint32_t def) {
        const char* v = getenv(name);
        return v ? (uint32_t)strtoul(v, nullptr, 10) : def;
    };
    ctx->tune_blocks_per_cu = std::max(1u, std::min(8u, env_u32("PTC_BLOCKS_PER_CU", ctx->tune_blocks_per_cu)));
    ctx->tune_refill = env_u32("PTC_REFILL", ctx->tune_refill);
    ctx->tune_refill_queued = env_u32("PTC_REFILL_QUEUED", getenv("PTC_REFILL") ? ctx->tune_refill : ctx->tune_refill_queued);     // (PTC_REFILL alone sets both)
    ctx->tune_ticket = env_u32("PTC_TICKET", ctx->tune_ticket);
    ctx->tune_wide_collapse = std::min(2u, env_u32("PTC_WIDE_COLLAPSE", ctx->tune_wide_collapse));
    ctx->tune_shade_blocks_per_cu = std::max(1u, env_u32("PTC_SHADE_BLOCKS_PER_CU", ctx->tune_shade_blocks_per_cu));
    ctx->tune_ring = env_u32("PTC_RING", ctx->tune_ring);
    ctx->tune_flatten = std::min(2u, env_u32("PTC_FLATTEN", ctx->tune_flatten));
    ctx->tune_trust_joins = env_u32("PTC_TRUST_JOINS", 0u) != 0u;
    ctx->tune_host_chunk = env_u32("PTC_HOST_CHUNK", 0u);
    if (ctx->tune_host_chunk) ctx->tune_host_chunk = std::max(1u << 12, std::min(1u << 28, ctx->tune_host_chunk));
    ctx->tune_host_slots = std::max(1u, std::min(PT_TRACE_SLOTS, env_u32("PTC_HOST_SLOTS", ctx->tune_host_slots)));
    ctx->tune_pass_paths = std::max(1u << 16, std::min(192u << 20, env_u32("PTC_PASS_PATHS", ctx->tune_pass_paths)));
    ctx->tune_work_limit = (size_t)env_u32("PTC_WORK_LIMIT_MB", 0u) << 20;
    ctx->tune_sample_group = std::max(1u, std::min(64u, env_u32("PTC_SAMPLE_GROUP", ctx->tune_sample_group)));
    while (ctx->tune_sample_group & (ctx->tune_sample_group - 1u)) ctx->tune_sample_group &= ctx->tune_sample_group - 1u;   // a power of two: k_accumulate cuts its 256 threads into groups
    ctx->tune_pass_slots = std::max(1u, std::min(4u, env_u32("PTC_PASS_SLOTS", ctx->tune_pass_slots)));
    ctx->tune_admit_frac = std::min(1.0f, (float)env_u32("PTC_ADMIT_PERCENT", (uint32_t)(ctx->tune_admit_frac * 100.0f + 0.5f)) / 100.0f);
    guard.p = nullptr;
    *out = ctx;
    return PT_OK;
}

extern "C" int pt_init(const int* device_ids, int n, pt_ctx** out) {
    return ptc_guard(nullptr, "pt_init", [&] { return pt_init_body(device_ids, n, out); });
}

static void ptc_slot_destroy(ptc_trace_slot& sl);

extern "C" void pt_shutdown(pt_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& sl : ctx->slots) ptc_slot_destroy(sl);
    if (ctx->spill) (void)hipFree(ctx->spill);
    if (ctx->work) (void)hipFree(ctx->work);
    if (ctx->io) (void)hipFree(ctx->io);
    if (ctx->redo) (void)hipFree(ctx->redo);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->ticket) (void)hipFree(ctx->ticket);
    for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->count_ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->trace_ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->pipe_ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->copy_in) { (void)hipStreamSynchronize(ctx->copy_in); (void)hipStreamDestroy(ctx->copy_in); }
    if (ctx->copy_out) { (void)hipStreamSynchronize(ctx->copy_out); (void)hipStreamDestroy(ctx->copy_out); }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int pt_set_traversal(pt_ctx* ctx, uint32_t mode) {
    if (!ctx) return PT_ERR_INVALID;
    if (mode != PT_TRAVERSE_MIN_T && mode != PT_TRAVERSE_EXACT && mode != PT_TRAVERSE_REPLAY) return ptc_fail(ctx, PT_ERR_INVALID, "unknown traversal mode");
    ctx->traversal = mode;
    return PT_OK;
}

extern "C" int pt_get_stats(const pt_ctx* ctx, pt_stats* out) {
    if (!ctx || !out) return PT_ERR_INVALID;
    *out = ctx->stats;
    return PT_OK;
}

extern "C" int pt_scene_get_info(const pt_scene* scene, pt_scene_info* out) {
    if (!scene || !out) return PT_ERR_INVALID;
    *out = scene->info;
    return PT_OK;
}

namespace {

// A host array whose pages are first touched by whoever fills it: a std::vector of 2.9 GB (20 M triangles) spends 0.5 s
// zeroing itself on one core before the worker threads overwrite every byte.
template <class T>
struct RawBuf {
    T* p = nullptr;
    size_t n = 0;
    explicit RawBuf(size_t count) : n(count) {
        if (n) { p = static_cast<T*>(std::malloc(n * sizeof(T))); if (!p) throw std::bad_alloc(); }
    }
    ~RawBuf() { std::free(p); }
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

template <class C, class T>
int upload(pt_ctx* ctx, pt_scene* sc, const C& host, const T** dev, const char* what) {
    *dev = nullptr;
    size_t bytes = host.size() * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);          // keep pointers valid for empty arrays
    void* p = nullptr;
    int rc = ptc_hip(ctx, hipMalloc(&p, bytes), what);
    if (rc != PT_OK) return rc;
    sc->allocations.push_back(p);
    sc->info.device_bytes += bytes;
    if (!host.empty()) {
        rc = ptc_hip(ctx, hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice), what);
        if (rc != PT_OK) return rc;
    }
    *dev = reinterpret_cast<const T*>(p);
    return PT_OK;
}

inline float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

int check_texture(pt_ctx* ctx, const pt_texture& t, const pt_scene_desc* d, uint32_t channels, const char* what) {
    if (t.kind == PT_TEX_FLAT) return PT_OK;
    if (t.kind != PT_TEX_NEAREST && t.kind != PT_TEX_BILINEAR) return ptc_fail(ctx, PT_ERR_INVALID, std::string(what) + ": unknown texture kind");
    if (t.image >= d->n_images || t.image >= PTC_IMG_BILINEAR) return ptc_fail(ctx, PT_ERR_INVALID, std::string(what) + ": image index out of range");
    if (d->images[t.image].channels != channels)
        return ptc_fail(ctx, PT_ERR_INVALID, std::string(what) + ": image has the wrong channel count");
    return PT_OK;
}

}  // namespace

extern "C" void pt_scene_destroy(pt_ctx* ctx, pt_scene* scene) {
    if (!scene) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        for (auto& sl : ctx->slots) if (sl.ready) (void)hipStreamSynchronize(sl.stream);      // batches in flight still read the scene
    }
    for (void* p : scene->allocations) (void)hipFree(p);
    delete scene;
}

extern "C" int pt_scene_create(pt_ctx* ctx, const pt_scene_desc* d, pt_scene** out) {
    return pt_scene_create_ex(ctx, d, 0u, out);
}

static int pt_scene_create_ex_body(pt_ctx* ctx, const pt_scene_desc* d, uint32_t flags, pt_scene** out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!d || !out) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: NULL argument");
    *out = nullptr;
    if (flags & ~PT_SCENE_BUILD_LBVH) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create_ex: unknown flag");
    const bool lbvh = (flags & PT_SCENE_BUILD_LBVH) != 0u;
    if (lbvh && (d->nodes || d->n_nodes)) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create_ex: PT_SCENE_BUILD_LBVH takes no nodes");
    const uint32_t N = d->n_triangles;
    if (N >= PTC_INDEX_MASK) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_scene_create: more than 2^30-2 triangles");
    if (N && !d->triangles) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: triangles is NULL");
    if (d->n_materials == 0 || !d->materials)
        return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: materials[0] (Material::default) is required");
    if (!lbvh && (N == 0 ? d->n_nodes != 0 : d->n_nodes != 2 * N - 1))
        return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: n_nodes must be 2*n_triangles-1 (BVH::build, one triangle per leaf)");
    if (d->n_nodes && !d->nodes) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: nodes is NULL");
    if (d->sky.kind != PT_SKY_D65 && d->sky.kind != PT_SKY_HOSEK_WILKIE)
        return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: unknown sky kind");
    if (d->sky.kind == PT_SKY_HOSEK_WILKIE && !(d->sky.turbidity >= 1.0f && d->sky.turbidity <= 10.0f))
        return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: turbidity outside [1, 10] (hosek_wilkie.rs:348-353)");
    PTC_HIP(ctx, hipSetDevice(ctx->device));

    // ---- validate + flatten the node vector ------------------------------------------------------
    // Every child index must point FORWARD (preorder) so that traversal terminates and never
    // leaves the array; every triangle must be referenced by exactly one leaf.  Done on the device after the uploads
    // (lbvh.hip: ptc_flatten_nodes) unless PTC_FLATTEN or PTC_WIDE_COLLAPSE ask for this host-side loop (1), or for both
    // and a byte-for-byte comparison (2).
    const uint32_t n_nodes = d->n_nodes;
    const bool host_flatten = !lbvh && (ctx->tune_flatten != 0u || ctx->tune_wide_collapse != 0u);
    std::vector<uint32_t> internal_rank(host_flatten ? n_nodes : 0u, 0u);
    uint32_t n_internal = 0;
    if (host_flatten) {
    for (uint32_t i = 0; i < n_nodes; ++i) {
        const pt_bvh_node& nd = d->nodes[i];
        if (nd.is_leaf) {
            if (nd.shape_index >= N) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: leaf shape_index out of range");
        } else {
            if (nd.child_l_index <= i || nd.child_l_index >= n_nodes || nd.child_r_index <= i || nd.child_r_index >= n_nodes)
                return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: child index is not a forward preorder index");
            internal_rank[i] = n_internal++;
        }
    }
    if (N && n_internal != N - 1) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: expected n_triangles-1 internal nodes");
    }
    std::vector<float4> nodes((size_t)n_internal * 4u);
    std::vector<uint32_t> rank_depth(n_internal, 0u);       // depth of internal node by rank (root = 1)
    uint32_t max_depth = 0;
    bool boxes_ordered = true;
    if (host_flatten) {
        // depth = internal nodes on the longest root-to-leaf path (bounds the traversal stack)
        std::vector<uint32_t> depth(n_nodes, 0u);
        if (n_nodes) depth[0] = d->nodes[0].is_leaf ? 0u : 1u;
        std::vector<uint8_t> seen(N, 0), reached(n_nodes, 0);
        for (uint32_t i = 0; i < n_nodes; ++i) {
            const pt_bvh_node& nd = d->nodes[i];
            if (i != 0 && !reached[i]) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: node unreachable from the root");
            if (nd.is_leaf) {
                if (seen[nd.shape_index]) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: triangle referenced by two leaves");
                seen[nd.shape_index] = 1;
                continue;
            }
            if (depth[i] > max_depth) max_depth = depth[i];
            rank_depth[internal_rank[i]] = depth[i];
            auto ref_of = [&](uint32_t child) -> uint32_t {
                const pt_bvh_node& c = d->nodes[child];
                reached[child] = 1;
                if (c.is_leaf) { depth[child] = depth[i]; return PTC_LEAF_BIT | c.shape_index; }
                depth[child] = depth[i] + 1;
                return internal_rank[child];
            };
            const uint32_t lref = ref_of(nd.child_l_index), rref = ref_of(nd.child_r_index);
            float4* q = &nodes[(size_t)internal_rank[i] * 4u];
            float lw, rw;
            std::memcpy(&lw, &lref, 4); std::memcpy(&rw, &rref, 4);
            for (int k = 0; k < 3; ++k) {       // an inverted box, or one with a NaN on one side only (never made by BVH::build), rules out the sign-select wide step
                const float a = nd.child_l_aabb.min[k], b = nd.child_l_aabb.max[k], c = nd.child_r_aabb.min[k], e = nd.child_r_aabb.max[k];
                if (!(a <= b || (a != a && b != b)) || !(c <= e || (c != c && e != e))) boxes_ordered = false;
            }
            q[0] = make_float4(nd.child_l_aabb.min[0], nd.child_l_aabb.min[1], nd.child_l_aabb.min[2], lw);
            q[1] = make_float4(nd.child_l_aabb.max[0], nd.child_l_aabb.max[1], nd.child_l_aabb.max[2], rw);
            q[2] = make_float4(nd.child_r_aabb.min[0], nd.child_r_aabb.min[1], nd.child_r_aabb.min[2], 0.0f);
            q[3] = make_float4(nd.child_r_aabb.max[0], nd.child_r_aabb.max[1], nd.child_r_aabb.max[2], 0.0f);
        }
    }
    if (max_depth > PTC_MAX_BVH_DEPTH) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_scene_create: BVH deeper than 4096 levels");

    // ---- wide nodes: collapse two binary levels into one record (device_types.h) -----------------------
    // Done on the device after the upload (lbvh.hip: depths, prefix sum, k_wide) unless PTC_WIDE_COLLAPSE says otherwise:
    // 1 = this host-side loop instead, 2 = both, and the two arrays must be byte-identical (consistency check for tests).
    std::vector<float4> wide;
    uint32_t wide_depth = 0;
    bool joins_exact = true;        // see DevSceneView::joins_exact (k_check_joins in lbvh.hip is the device's form of the loop below)
    const uint32_t wide_mode = lbvh ? 0u : ctx->tune_wide_collapse;
    if (n_internal && wide_mode != 0u) {
        // wide nodes = internal nodes at odd depth, numbered in node (pre)order; slots = the grandchildren at fixed
        // places in REVERSE binary order (RR, RL | LR, LL; a leaf child takes the first slot of its pair, the other stays
        // a NaN box): device_types.h.  Same rule as k_wide in lbvh.hip.
        const float nan = std::numeric_limits<float>::quiet_NaN();
        std::vector<uint32_t> wide_index(n_internal);
        uint32_t n_wide = 0;
        for (uint32_t r = 0; r < n_internal; ++r) { wide_index[r] = n_wide; n_wide += rank_depth[r] & 1u; }
        wide.assign((size_t)n_wide * 8u, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        wide_depth = (max_depth + 1u) / 2u;
        for (uint32_t r = 0; r < n_internal; ++r) {
            if (!(rank_depth[r] & 1u)) continue;
            float lo[3][4], hi[3][4];
            uint32_t refs[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            for (int a = 0; a < 3; ++a) for (int k = 0; k < 4; ++k) { lo[a][k] = nan; hi[a][k] = nan; }
            auto take = [&](const float4* q, int side, int slot) {
                const float4 l = q[side ? 2 : 0], h = q[side ? 3 : 1];
                lo[0][slot] = l.x; lo[1][slot] = l.y; lo[2][slot] = l.z; hi[0][slot] = h.x; hi[1][slot] = h.y; hi[2][slot] = h.z;
                uint32_t ref; std::memcpy(&ref, side ? &q[1].w : &q[0].w, 4);
                refs[slot] = (ref & PTC_LEAF_BIT) ? ref : wide_index[ref];
            };
            const float4* q = &nodes[(size_t)r * 4u];
            for (int side = 1; side >= 0; --side) {          // slots 0, 1: the right child's pair; 2, 3: the left child's
                const int base = side ? 0 : 2;
                uint32_t ref; std::memcpy(&ref, side ? &q[1].w : &q[0].w, 4);
                if (ref & PTC_LEAF_BIT) take(q, side, base);
                else {
                    const float4* q2 = &nodes[(size_t)ref * 4u];
                    take(q2, 1, base); take(q2, 0, base + 1);
                    const float4 slo = q[side ? 2 : 0], shi = q[side ? 3 : 1];
                    const float sl[3] = {slo.x, slo.y, slo.z}, sh[3] = {shi.x, shi.y, shi.z};
                    const float jl[3] = {pt_fmin(q2[0].x, q2[2].x), pt_fmin(q2[0].y, q2[2].y), pt_fmin(q2[0].z, q2[2].z)};
                    const float jh[3] = {pt_fmax(q2[1].x, q2[3].x), pt_fmax(q2[1].y, q2[3].y), pt_fmax(q2[1].z, q2[3].z)};
                    for (int k = 0; k < 3; ++k)
                        if (!(sl[k] == jl[k] || (sl[k] != sl[k] && jl[k] != jl[k])) || !(sh[k] == jh[k] || (sh[k] != sh[k] && jh[k] != jh[k]))) joins_exact = false;
                }
            }
            float4* w = &wide[(size_t)wide_index[r] * 8u];
            for (int a = 0; a < 3; ++a) {
                w[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
                w[3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
            }
            float rb[4];
            std::memcpy(rb, refs, 16);
            w[6] = make_float4(rb[0], rb[1], rb[2], rb[3]);
        }
    }

    // ---- Triangle::new (triangle.rs:50-60) ---------------------------------------------------------
    RawBuf<float4> tris((size_t)N * PTC_TRI_STRIDE), shade((size_t)N * 5u), boxes(lbvh ? (size_t)N * 2u : 0u);   // every entry is written below
    std::atomic<bool> bad_material{false};
    auto precompute = [&](uint32_t begin, uint32_t end) {
    for (uint32_t i = begin; i < end; ++i) {
        const pt_triangle& t = d->triangles[i];
        if (lbvh) {          // Triangle::aabb (triangle.rs:118-122): grow by a, b, c
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) {
                lo[k] = pt_fmin(pt_fmin(pt_fmin(pt_inf(), t.a[k]), t.b[k]), t.c[k]);
                hi[k] = pt_fmax(pt_fmax(pt_fmax(-pt_inf(), t.a[k]), t.b[k]), t.c[k]);
            }
            boxes[2 * (size_t)i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
            boxes[2 * (size_t)i + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        }
        if (t.material_index >= d->n_materials || t.material_index >= (1u << 30)) { bad_material = true; return; }
        float ab[3], ac[3], n[3], n1[3], n2[3], tmp[3];
        for (int k = 0; k < 3; ++k) { ab[k] = t.b[k] - t.a[k]; ac[k] = t.c[k] - t.a[k]; }
        cross3(ab, ac, n);
        const float dd = dot3(t.a, n);
        const float nn = dot3(n, n);
        cross3(ac, n, tmp);
        for (int k = 0; k < 3; ++k) n1[k] = tmp[k] / nn;
        const float d1 = -dot3(t.a, n1);
        cross3(n, ab, tmp);
        for (int k = 0; k < 3; ++k) n2[k] = tmp[k] / nn;
        const float d2 = -dot3(t.a, n2);
        tris[(size_t)i * PTC_TRI_STRIDE + 0] = make_float4(n[0], n[1], n[2], dd);
        tris[(size_t)i * PTC_TRI_STRIDE + 1] = make_float4(n1[0], n1[1], n1[2], d1);
        tris[(size_t)i * PTC_TRI_STRIDE + 2] = make_float4(n2[0], n2[1], n2[2], d2);
        float mw, fw;
        const uint32_t flags = t.flags & (PT_TRI_HAS_UV | PT_TRI_HAS_NORMALS);
        std::memcpy(&mw, &t.material_index, 4); std::memcpy(&fw, &flags, 4);
        // what every scatter needs, in the fourth quarter of the record: Intersection.normal (= n) and material | flags
        const uint32_t mf = (t.material_index << 2) | flags;
        float mfw;
        std::memcpy(&mfw, &mf, 4);
        tris[(size_t)i * PTC_TRI_STRIDE + 3] = make_float4(n[0], n[1], n[2], mfw);
        float4* s = &shade[(size_t)i * 5];
        s[0] = make_float4(mw, fw, t.uv[0][0], t.uv[0][1]);
        s[1] = make_float4(t.uv[1][0], t.uv[1][1], t.uv[2][0], t.uv[2][1]);
        s[2] = make_float4(t.vertex_normals[0][0], t.vertex_normals[0][1], t.vertex_normals[0][2], t.vertex_normals[1][0]);
        s[3] = make_float4(t.vertex_normals[1][1], t.vertex_normals[1][2], t.vertex_normals[2][0], t.vertex_normals[2][1]);
        s[4] = make_float4(t.vertex_normals[2][2], 0.0f, 0.0f, 0.0f);
    }
    };
    {   // every triangle is independent: split the range over the host's cores (at most 16)
        const uint32_t nt = N < (1u << 16) ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        const uint32_t chunk = (N + nt - 1u) / nt;
        for (uint32_t k = 1; k < nt; ++k)
            if (k * chunk < N) {
                try { pool.emplace_back(precompute, k * chunk, std::min(N, (k + 1u) * chunk)); }
                catch (...) { precompute(k * chunk, std::min(N, (k + 1u) * chunk)); }       // no thread to be had: do it here
            }
        precompute(0u, std::min(N, chunk));
        for (auto& th : pool) th.join();
    }
    if (bad_material) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: material_index out of range");

    // ---- materials, images ---------------------------------------------------------------------------
    std::vector<DevImage> images(d->n_images);
    std::vector<float> texels;
    for (uint32_t i = 0; i < d->n_images; ++i) {
        const pt_image& im = d->images[i];
        if (!im.texels || im.width == 0 || im.height == 0 || (im.channels != 1 && im.channels != 3))
            return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: bad image");
        images[i] = DevImage{im.width, im.height, im.channels, (uint32_t)texels.size()};
        texels.insert(texels.end(), im.texels, im.texels + (size_t)im.width * im.height * im.channels);
    }
    std::vector<DevMaterial> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const pt_material& m = d->materials[i];
        int rc;
        if ((rc = check_texture(ctx, m.diffuse, d, 3, "pt_scene_create: material.diffuse")) != PT_OK) return rc;
        if ((rc = check_texture(ctx, m.dissolve, d, 1, "pt_scene_create: material.dissolve")) != PT_OK) return rc;
        if (m.emit_kind != PT_EMIT_NULL && m.emit_kind != PT_EMIT_ABSORBED_D65)
            return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_create: unknown emit kind");
        if (m.emit_kind == PT_EMIT_ABSORBED_D65 && (rc = check_texture(ctx, m.absorption, d, 3, "pt_scene_create: material.absorption")) != PT_OK) return rc;
        DevMaterial& o = mats[i];
        for (int k = 0; k < 3; ++k) { o.diffuse[k] = m.diffuse.value[k]; o.absorb[k] = m.absorption.value[k]; }
        // image reference of a texture: 0xFFFFFFFF = Flat, else the image index, bit 31 set for Bilinear (DevMaterial)
        auto img_ref = [](const pt_texture& t) -> uint32_t {
            return t.kind == PT_TEX_NEAREST ? t.image : t.kind == PT_TEX_BILINEAR ? (t.image | PTC_IMG_BILINEAR) : 0xFFFFFFFFu;
        };
        o.diffuse_img = img_ref(m.diffuse);
        o.absorb_img = m.emit_kind == PT_EMIT_ABSORBED_D65 ? img_ref(m.absorption) : 0xFFFFFFFFu;
        o.base_dissolve = m.base_dissolve;
        o.dissolve_value = m.dissolve.value[0];
        o.dissolve_img = img_ref(m.dissolve);
        o.emit_kind = m.emit_kind;
    }

    // ---- sky ------------------------------------------------------------------------------------------
    auto sc = std::make_unique<pt_scene>();
    DevSky& sky = sc->sky;
    std::memset(&sky, 0, sizeof(sky));
    sky.kind = d->sky.kind; sky.power = d->sky.power;
    sky.turbidity = d->sky.turbidity; sky.solar_radius = d->sky.solar_radius;
    std::memcpy(sky.zenith, d->sky.zenith, sizeof(sky.zenith));
    std::memcpy(sky.sun, d->sky.sun, sizeof(sky.sun));
    std::memcpy(sky.configs, d->sky.configs, sizeof(sky.configs));
    std::memcpy(sky.radiances, d->sky.radiances, sizeof(sky.radiances));
    std::vector<float> solar(11 * 2 * 45 * 4, 0.0f), limb(ptc_tab_hw_limb, ptc_tab_hw_limb + 66);
    if (sky.kind == PT_SKY_HOSEK_WILKIE) {
        uint32_t tl = pt_f2usize(sky.turbidity) - 1u;                         // hosek_wilkie.rs:355
        if (tl > 8u) tl = 8u;
        sky.turb_low = tl;
        sky.turb_frac = sky.turbidity - (float)(tl + 1u);                     // :356
        for (int wl = 0; wl < 11; ++wl)
            for (int k = 0; k < 2; ++k)
                std::memcpy(&solar[((size_t)wl * 2 + k) * 180], &ptc_tab_hw_solar[((size_t)wl * 10 + tl + k) * 180],
                            180 * sizeof(float));
    }
    std::vector<float> refl(ptc_tab_srgb2refl, ptc_tab_srgb2refl + 471 * 3), d65(ptc_tab_d65, ptc_tab_d65 + 531);
    std::vector<DevSky> skyv(1, sky);

    // ---- upload ---------------------------------------------------------------------------------------
    pt_scene* s = sc.get();
    int rc;
    const DevSky* d_sky = nullptr;
    if ((rc = upload(ctx, s, nodes, &s->view.nodes, "nodes")) != PT_OK ||
        (rc = upload(ctx, s, wide, &s->view.wide, "wide nodes")) != PT_OK ||
        (rc = upload(ctx, s, tris, &s->view.tris, "triangles")) != PT_OK ||
        (rc = upload(ctx, s, shade, &s->view.shade, "shading records")) != PT_OK ||
        (rc = upload(ctx, s, mats, &s->view.materials, "materials")) != PT_OK ||
        (rc = upload(ctx, s, images, &s->view.images, "images")) != PT_OK ||
        (rc = upload(ctx, s, texels, &s->view.texels, "texels")) != PT_OK ||
        (rc = upload(ctx, s, refl, &s->view.refl_tab, "reflectance table")) != PT_OK ||
        (rc = upload(ctx, s, d65, &s->view.d65_tab, "d65 table")) != PT_OK ||
        (rc = upload(ctx, s, solar, &s->view.solar_tab, "solar table")) != PT_OK ||
        (rc = upload(ctx, s, limb, &s->view.limb_tab, "limb table")) != PT_OK ||
        (rc = upload(ctx, s, skyv, &d_sky, "sky")) != PT_OK) {
        pt_scene_destroy(ctx, sc.release());
        return rc;
    }
    ctx->stats = pt_stats{};
    if (!lbvh && (!host_flatten || ctx->tune_flatten == 2u)) {
        float4* d_nodes = nullptr;
        uint32_t dev_internal = 0;
        bool dev_ordered = true;
        rc = ptc_flatten_nodes(ctx, d->nodes, n_nodes, N, &d_nodes, &dev_internal, &dev_ordered);
        if (rc != PT_OK) { pt_scene_destroy(ctx, sc.release()); return rc; }
        if (d_nodes) s->allocations.push_back(d_nodes);
        if (host_flatten) {                              // PTC_FLATTEN=2: the two must agree on every byte
            std::vector<float4> back((size_t)dev_internal * 4u);
            const bool same = dev_internal == n_internal && dev_ordered == boxes_ordered &&
                              (back.empty() || (hipMemcpy(back.data(), d_nodes, back.size() * sizeof(float4), hipMemcpyDeviceToHost) == hipSuccess &&
                                                std::memcmp(back.data(), nodes.data(), back.size() * sizeof(float4)) == 0));
            if (!same) { pt_scene_destroy(ctx, sc.release()); return ptc_fail(ctx, PT_ERR_HIP, "pt_scene_create: device-side node flattening disagrees with the host's"); }
        } else {
            s->view.nodes = d_nodes;
            s->info.device_bytes += (size_t)dev_internal * 64u;
            n_internal = dev_internal;
            boxes_ordered = dev_ordered;
        }
    }
    if (!lbvh && n_internal && wide_mode != 1u) {
        float4* d_wide = nullptr;
        uint32_t n_wide = 0, dev_depth = 0, dev_wide_depth = 0;
        bool dev_joins = true;
        rc = ptc_build_wide_from_nodes(ctx, s->view.nodes, n_internal, &d_wide, &n_wide, &dev_depth, &dev_wide_depth, &dev_joins);
        if (rc != PT_OK) { pt_scene_destroy(ctx, sc.release()); return rc; }
        s->allocations.push_back(d_wide);
        if (!host_flatten) max_depth = dev_depth;
        bool same = dev_depth == max_depth && (wide_mode != 2u || dev_joins == joins_exact);
        joins_exact = dev_joins;
        if (wide_mode == 2u) {
            std::vector<float4> back((size_t)n_wide * 8u);
            same = same && back.size() == wide.size() && dev_wide_depth == wide_depth &&
                   hipMemcpy(back.data(), d_wide, back.size() * sizeof(float4), hipMemcpyDeviceToHost) == hipSuccess &&
                   std::memcmp(back.data(), wide.data(), back.size() * sizeof(float4)) == 0;
        }
        if (!same) { pt_scene_destroy(ctx, sc.release()); return ptc_fail(ctx, PT_ERR_HIP, "pt_scene_create: device-side wide collapse disagrees with the host's"); }
        s->view.wide = d_wide;
        s->info.device_bytes += (size_t)n_wide * 128u;
        wide_depth = dev_wide_depth;
    }
    if (lbvh && N >= 2) {
        float4 *d_nodes = nullptr, *d_wide = nullptr;
        uint32_t n_wide = 0;
        float ms = 0.0f;
        rc = ptc_build_lbvh(ctx, boxes.data(), N, &d_nodes, &d_wide, &n_wide, &max_depth, &wide_depth, &ms);
        if (rc != PT_OK) { pt_scene_destroy(ctx, sc.release()); return rc; }
        s->allocations.push_back(d_nodes); s->allocations.push_back(d_wide);
        s->view.nodes = d_nodes; s->view.wide = d_wide;
        s->info.device_bytes += (size_t)(N - 1) * 64u + (size_t)n_wide * 128u;
        n_internal = N - 1;
        ctx->stats.total_ms = ms;
    }
    s->d_sky = const_cast<DevSky*>(d_sky);
    s->view.n_tris = N;
    s->view.n_materials = d->n_materials;
    s->view.max_depth = max_depth;
    s->view.wide_depth = wide_depth;
    s->view.sign_select_ok = (boxes_ordered && (uint64_t)n_internal * 128u < (1ull << 32)) ? 1u : 0u;      // (n_wide <= n_internal)
    s->view.joins_exact = (joins_exact || ctx->tune_trust_joins) ? 1u : 0u;
    s->view.root_ref = (N == 1) ? (PTC_LEAF_BIT | (lbvh ? 0u : d->nodes[0].shape_index)) : 0u;
    s->info.n_triangles = N;
    s->info.n_internal_nodes = n_internal;
    s->info.max_depth = max_depth;
    *out = sc.release();
    return PT_OK;
}

extern "C" int pt_scene_create_ex(pt_ctx* ctx, const pt_scene_desc* d, uint32_t flags, pt_scene** out) {
    return ptc_guard(ctx, "pt_scene_create_ex", [&] { return pt_scene_create_ex_body(ctx, d, flags, out); });
}

// The scene's binary tree back in the reference's form (enum BVHNode, bvh.rs:123-127, in `BVH.nodes` order: preorder,
// leaves as entries of their own).  For a scene made from the caller's nodes this is the caller's vector again; for a
// device-built tree (PT_SCENE_BUILD_LBVH) it is what a host needs to hold that tree as a `BVH { nodes }` of its own.
static int pt_scene_get_bvh_body(pt_ctx* ctx, const pt_scene* scene, pt_bvh_node* nodes_out, uint32_t capacity,
                                uint32_t* n_nodes_out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!scene || !n_nodes_out) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_get_bvh: NULL argument");
    const uint32_t N = scene->view.n_tris;
    const uint32_t total = N ? 2u * N - 1u : 0u;
    *n_nodes_out = total;
    if (!nodes_out) return PT_OK;                    // size query
    if (capacity < total) return ptc_fail(ctx, PT_ERR_INVALID, "pt_scene_get_bvh: nodes_out holds fewer than 2 * n_triangles - 1 entries");
    if (total == 0) return PT_OK;
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<float4> dev;
    try { dev.resize((size_t)(N - 1u) * 4u); } catch (const std::bad_alloc&) { return ptc_fail(ctx, PT_ERR_NOMEM, "pt_scene_get_bvh: out of host memory"); }
    if (N > 1u) PTC_HIP(ctx, hipMemcpy(dev.data(), scene->view.nodes, dev.size() * sizeof(float4), hipMemcpyDeviceToHost));
    struct Item { uint32_t ref, parent; bool right; };
    std::vector<Item> todo;
    todo.push_back({scene->view.root_ref, 0xFFFFFFFFu, false});
    uint32_t count = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (count == total) return ptc_fail(ctx, PT_ERR_HIP, "pt_scene_get_bvh: the device tree is not a tree over the scene's triangles");
        const uint32_t idx = count++;
        pt_bvh_node& o = nodes_out[idx];
        std::memset(&o, 0, sizeof(o));
        if (it.parent != 0xFFFFFFFFu) (it.right ? nodes_out[it.parent].child_r_index : nodes_out[it.parent].child_l_index) = idx;
        if (it.ref & PTC_LEAF_BIT) {
            o.is_leaf = 1u;
            o.shape_index = it.ref & PTC_INDEX_MASK;
            continue;
        }
        if (it.ref >= N - 1u) return ptc_fail(ctx, PT_ERR_HIP, "pt_scene_get_bvh: node reference out of range");
        const float4* q = &dev[(size_t)it.ref * 4u];
        o.child_l_aabb = pt_aabb{{q[0].x, q[0].y, q[0].z}, {q[1].x, q[1].y, q[1].z}};
        o.child_r_aabb = pt_aabb{{q[2].x, q[2].y, q[2].z}, {q[3].x, q[3].y, q[3].z}};
        uint32_t lref, rref;
        std::memcpy(&lref, &q[0].w, 4);
        std::memcpy(&rref, &q[1].w, 4);
        todo.push_back({rref, idx, true});
        todo.push_back({lref, idx, false});           // popped first: parent, left subtree, right subtree (bvh.rs:163-164)
    }
    if (count != total) return ptc_fail(ctx, PT_ERR_HIP, "pt_scene_get_bvh: the device tree is not a tree over the scene's triangles");
    return PT_OK;
}

extern "C" int pt_scene_get_bvh(pt_ctx* ctx, const pt_scene* scene, pt_bvh_node* nodes_out, uint32_t capacity,
                                uint32_t* n_nodes_out) {
    return ptc_guard(ctx, "pt_scene_get_bvh", [&] { return pt_scene_get_bvh_body(ctx, scene, nodes_out, capacity, n_nodes_out); });
}

// ---- seam 2: closest hit ------------------------------------------------------------------------------

static int pt_trace_closest_dev_body(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                                    const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!scene || (n && (!d_rays || !d_out))) return ptc_fail(ctx, PT_ERR_INVALID, "pt_trace_closest_dev: NULL argument");
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    ctx->stats = pt_stats{};
    if (n == 0) return PT_OK;
    PTC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    int rc = ptc_launch_trace(ctx, scene, d_rays, d_start_tri, nullptr, nullptr, n, d_out);
    if (rc != PT_OK) return rc;
    PTC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    PTC_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    float ms = 0.0f;
    PTC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->stats.trace_ms = ms; ctx->stats.total_ms = ms;
    ctx->stats.trace_rays = n; ctx->stats.trace_launches = 1;
    return PT_OK;
}

extern "C" int pt_trace_closest_dev(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                                    const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out) {
    return ptc_guard(ctx, "pt_trace_closest_dev", [&] { return pt_trace_closest_dev_body(ctx, scene, d_rays, d_start_tri, n, d_out); });
}

// ---- closest hit without the wait: PT_TRACE_SLOTS slots, each a stream and traversal scratch of its own ------------------
static int ptc_slot_setup(pt_ctx* ctx, ptc_trace_slot& sl) {
    if (sl.ready) return PT_OK;
    hipStream_t st = nullptr;
    PTC_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t* tickets = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void**)&tickets, PTC_TICKET_BYTES);
    if (e == hipSuccess) e = hipMemsetAsync(tickets, 0, PTC_TICKET_BYTES, st);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e != hipSuccess) {
        if (e1) (void)hipEventDestroy(e1);
        if (e0) (void)hipEventDestroy(e0);
        (void)hipStreamSynchronize(st);
        if (tickets) (void)hipFree(tickets);
        (void)hipStreamDestroy(st);
        return ptc_hip(ctx, e, "pt_trace_closest_dev_submit: slot set-up");
    }
    sl.ready = true;
    sl.stream = st;
    sl.ev_begin = e0; sl.ev_end = e1;
    sl.s = ptc_trace_scratch{};
    sl.s.tickets = tickets;
    sl.s.launch_parity = &sl.launch_parity;
    sl.s.redo_parity = &sl.redo_parity;
    sl.s.stream = st;
    sl.s.grow = true;
    return PT_OK;
}

static void ptc_slot_destroy(ptc_trace_slot& sl) {
    if (!sl.ready) return;
    (void)hipStreamSynchronize(sl.stream);
    if (sl.s.spill) (void)hipFree(sl.s.spill);
    if (sl.s.redo) (void)hipFree(sl.s.redo);
    if (sl.s.tickets) (void)hipFree(sl.s.tickets);
    if (sl.ev_begin) (void)hipEventDestroy(sl.ev_begin);
    if (sl.ev_end) (void)hipEventDestroy(sl.ev_end);
    (void)hipStreamDestroy(sl.stream);
    sl = ptc_trace_slot{};
}

static int pt_trace_closest_dev_submit_body(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                                           const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out, uint32_t slot) {
    if (!ctx) return PT_ERR_INVALID;
    if (!scene || (n && (!d_rays || !d_out))) return ptc_fail(ctx, PT_ERR_INVALID, "pt_trace_closest_dev_submit: NULL argument");
    if (slot >= PT_TRACE_SLOTS) return ptc_fail(ctx, PT_ERR_INVALID, "pt_trace_closest_dev_submit: slot out of range (PT_TRACE_SLOTS)");
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) return PT_OK;
    ptc_trace_slot& sl = ctx->slots[slot];
    int rc = ptc_slot_setup(ctx, sl);
    if (rc != PT_OK) return rc;
    if (sl.pending == 0u) PTC_HIP(ctx, hipEventRecord(sl.ev_begin, sl.stream));
    if ((rc = ptc_launch_trace_in(ctx, scene, d_rays, d_start_tri, nullptr, nullptr, n, d_out, &sl.s)) != PT_OK) return rc;
    PTC_HIP(ctx, hipEventRecord(sl.ev_end, sl.stream));
    sl.pending += 1u;
    sl.rays += n;
    return PT_OK;
}

extern "C" int pt_trace_closest_dev_submit(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                                           const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out, uint32_t slot) {
    return ptc_guard(ctx, "pt_trace_closest_dev_submit", [&] { return pt_trace_closest_dev_submit_body(ctx, scene, d_rays, d_start_tri, n, d_out, slot); });
}

static int pt_trace_closest_dev_wait_body(pt_ctx* ctx, uint32_t slot) {
    if (!ctx) return PT_ERR_INVALID;
    if (slot >= PT_TRACE_SLOTS) return ptc_fail(ctx, PT_ERR_INVALID, "pt_trace_closest_dev_wait: slot out of range (PT_TRACE_SLOTS)");
    ctx->stats = pt_stats{};
    ptc_trace_slot& sl = ctx->slots[slot];
    if (!sl.ready || sl.pending == 0u) return PT_OK;
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t launches = sl.pending;
    const uint64_t rays = sl.rays;
    sl.pending = 0u; sl.rays = 0u;          // whatever happens below, the slot starts afresh
    PTC_HIP(ctx, hipStreamSynchronize(sl.stream));
    float ms = 0.0f;
    PTC_HIP(ctx, hipEventElapsedTime(&ms, sl.ev_begin, sl.ev_end));
    ctx->stats.trace_ms = ms; ctx->stats.total_ms = ms;
    ctx->stats.trace_rays = rays; ctx->stats.trace_launches = launches;
    return PT_OK;
}

extern "C" int pt_trace_closest_dev_wait(pt_ctx* ctx, uint32_t slot) {
    return ptc_guard(ctx, "pt_trace_closest_dev_wait", [&] { return pt_trace_closest_dev_wait_body(ctx, slot); });
}

// pinned host memory for callers that want the host-pointer entry points to run at PCIe speed
extern "C" int pt_host_alloc(size_t bytes, void** out) {
    if (!out) return PT_ERR_INVALID;
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return ptc_fail(nullptr, e == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_HIP, std::string("pt_host_alloc: ") + hipGetErrorString(e));
    return PT_OK;
}
extern "C" void pt_host_free(void* p) { if (p) (void)hipHostFree(p); }

static int trace_closest_host(pt_ctx* ctx, const pt_scene* scene, const pt_ray* rays, const uint32_t* start_tri, uint32_t n, pt_hit* out) {
    const size_t rb = (size_t)n * sizeof(pt_ray), sb = (size_t)n * sizeof(uint32_t), hb = (size_t)n * sizeof(pt_hit);
    const size_t sb_pad = (sb + 255) & ~(size_t)255;
    int rc = ptc_reserve(ctx, &ctx->io, &ctx->io_bytes, rb + sb_pad + hb, "ray staging");
    if (rc != PT_OK) return rc;
    char* base = (char*)ctx->io;
    pt_ray* d_rays = (pt_ray*)base;
    uint32_t* d_start = (uint32_t*)(base + rb);
    pt_hit* d_out = (pt_hit*)(base + rb + sb_pad);
    // Chunks of PTC_HOST_CHUNK rays: the upload of chunk c+1, the kernel of chunk c and the download of chunk c-1 overlap
    // (streams chained by events).  The kernels of consecutive chunks go to alternating trace slots (PTC_HOST_SLOTS of them, the
    // streams and scratch of pt_trace_closest_dev_submit), so that the drain of one chunk's launch fills with the head of the
    // next.  With pinned host buffers (pt_host_alloc) every copy is asynchronous and the call runs at the speed of the slower
    // PCIe direction; pageable buffers make the runtime stage each copy, which serialises them on the calling thread (the
    // kernels still overlap).
    // (chunk size: PTC_HOST_CHUNK when set; else 512 Ki rays when the caller's rays are in pinned memory - the copies are DMA, smaller
    // chunks start the first kernel sooner and two slots hide their launches' drains: 2.60 -> 2.12 ms per 2 M rays - and 1 Mi rays
    // for pageable memory, where every chunk costs a staged copy on the calling thread)
    uint32_t CHUNK = ctx->tune_host_chunk;
    if (CHUNK == 0u) {
        hipPointerAttribute_t at{};
        const bool pinned = hipPointerGetAttributes(&at, rays) == hipSuccess && at.type == hipMemoryTypeHost;
        (void)hipGetLastError();          // an unregistered pointer is reported as an error by some runtimes: not ours to keep
        CHUNK = pinned ? (1u << 19) : (1u << 20);
    }
    const uint32_t n_chunks = (n + CHUNK - 1u) / CHUNK;
    const uint32_t n_slots = std::min<uint32_t>(std::min<uint32_t>(ctx->tune_host_slots, PT_TRACE_SLOTS), n_chunks);
    for (uint32_t k = 0; k < n_slots; ++k)
        if ((rc = ptc_slot_setup(ctx, ctx->slots[k])) != PT_OK) return rc;
    // per chunk: {rays uploaded, hits ready} without timing, {kernel start, kernel end} with
    while (ctx->pipe_ev.size() < 2u * (size_t)n_chunks) {
        hipEvent_t e = nullptr;
        PTC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pipe_ev.push_back(e);
    }
    while (ctx->trace_ev.size() < 2u * (size_t)n_chunks) {
        hipEvent_t e = nullptr;
        PTC_HIP(ctx, hipEventCreate(&e));
        ctx->trace_ev.push_back(e);
    }
    ctx->stats = pt_stats{};
    PTC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    for (uint32_t k = 0; k < n_slots; ++k) PTC_HIP(ctx, hipStreamWaitEvent(ctx->slots[k].stream, ctx->ev[0], 0));
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const uint32_t b = c * CHUNK, m = std::min(CHUNK, n - b);
        ptc_trace_slot& sl = ctx->slots[c % n_slots];
        PTC_HIP(ctx, hipMemcpyAsync(d_rays + b, rays + b, (size_t)m * sizeof(pt_ray), hipMemcpyHostToDevice, ctx->copy_in));
        if (start_tri) PTC_HIP(ctx, hipMemcpyAsync(d_start + b, start_tri + b, (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->copy_in));
        PTC_HIP(ctx, hipEventRecord(ctx->pipe_ev[2u * c], ctx->copy_in));
        PTC_HIP(ctx, hipStreamWaitEvent(sl.stream, ctx->pipe_ev[2u * c], 0));
        PTC_HIP(ctx, hipEventRecord(ctx->trace_ev[2u * c], sl.stream));
        if ((rc = ptc_launch_trace_in(ctx, scene, d_rays + b, start_tri ? d_start + b : nullptr, nullptr, nullptr, m, d_out + b, &sl.s)) != PT_OK) return rc;
        PTC_HIP(ctx, hipEventRecord(ctx->trace_ev[2u * c + 1u], sl.stream));
        PTC_HIP(ctx, hipEventRecord(ctx->pipe_ev[2u * c + 1u], sl.stream));
        PTC_HIP(ctx, hipStreamWaitEvent(ctx->copy_out, ctx->pipe_ev[2u * c + 1u], 0));
        PTC_HIP(ctx, hipMemcpyAsync(out + b, d_out + b, (size_t)m * sizeof(pt_hit), hipMemcpyDeviceToHost, ctx->copy_out));
    }
    for (uint32_t c = n_chunks - n_slots; c < n_chunks; ++c) PTC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_ev[2u * c + 1u], 0));
    PTC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    PTC_HIP(ctx, hipStreamSynchronize(ctx->copy_out));
    PTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // kernels of different slots overlap: their time is first start to last end
    float trace_ms = 0.0f;
    for (uint32_t c = n_chunks - n_slots; c < n_chunks; ++c) {
        float ms = 0.0f;
        PTC_HIP(ctx, hipEventElapsedTime(&ms, ctx->trace_ev[0], ctx->trace_ev[2u * c + 1u]));
        trace_ms = std::max(trace_ms, ms);
    }
    float total = 0.0f;
    PTC_HIP(ctx, hipEventElapsedTime(&total, ctx->ev[0], ctx->ev[1]));
    ctx->stats.trace_ms = trace_ms; ctx->stats.total_ms = total;
    ctx->stats.trace_rays = n; ctx->stats.trace_launches = n_chunks;
    return PT_OK;
}

static int pt_trace_closest_body(pt_ctx* ctx, const pt_scene* scene, const pt_ray* rays,
                                const uint32_t* start_tri, uint32_t n, pt_hit* out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!scene || (n && (!rays || !out))) return ptc_fail(ctx, PT_ERR_INVALID, "pt_trace_closest: NULL argument");
    if (n == 0) { ctx->stats = pt_stats{}; return PT_OK; }
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = trace_closest_host(ctx, scene, rays, start_tri, n, out);
    if (rc != PT_OK) {      // leave nothing in flight that still reads or writes the caller's buffers
        (void)hipStreamSynchronize(ctx->copy_in); (void)hipStreamSynchronize(ctx->stream);
        for (auto& sl : ctx->slots) if (sl.ready) (void)hipStreamSynchronize(sl.stream);
        (void)hipStreamSynchronize(ctx->copy_out);
    }
    return rc;
}

extern "C" int pt_trace_closest(pt_ctx* ctx, const pt_scene* scene, const pt_ray* rays,
                                const uint32_t* start_tri, uint32_t n, pt_hit* out) {
    return ptc_guard(ctx, "pt_trace_closest", [&] { return pt_trace_closest_body(ctx, scene, rays, start_tri, n, out); });
}
