// device_types.h — HBM layouts of the render core and the context/scene objects behind the C ABI.
#ifndef PTC_DEVICE_TYPES_H
#define PTC_DEVICE_TYPES_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ptcore.h"

// ---- child references -------------------------------------------------------------------------
// bit 31: leaf (low bits = triangle index), else low bits = internal-node index.
// bit 30 is reserved for the EXACT traversal's frame tag, so indices are < 2^30.
#define PTC_LEAF_BIT  0x80000000u
#define PTC_FRAME_BIT 0x40000000u
#define PTC_INDEX_MASK 0x3FFFFFFFu
#define PTC_NONE_ID   0x3FFFFFFFu      /* "no triangle" inside a tagged frame */

// ---- BVH node: 64 B = 4 x float4, internal nodes only (N-1 of them), preorder among internals.
// The boxes of both children live in the parent, as in enum BVHNode::Node (bvh.rs:126).
//   q0 = { l.min.x, l.min.y, l.min.z, bits(left ref)  }
//   q1 = { l.max.x, l.max.y, l.max.z, bits(right ref) }
//   q2 = { r.min.x, r.min.y, r.min.z, 0 }
//   q3 = { r.max.x, r.max.y, r.max.z, 0 }
// ---- wide node (PT_TRAVERSE_EXACT, PT_TRAVERSE_MIN_T): 128 B = 8 x float4, two levels of the SAME binary tree in one
// record: the slots of a wide node are the grandchildren of a binary node X, in REVERSE binary order and at FIXED places:
// slots 0, 1 = the right child's right and left child (RR, RL), slots 2, 3 = the left child's (LR, LL).  A child of X that
// is a leaf sits in the first slot of its pair (0 or 2) with its own box; the other slot of the pair is empty: NaN box
// (never hit, and dropped by minNum / maxNum when the child's box is re-derived from its pair: trace_step.h), ref
// 0xFFFFFFFF.  Every box and every triangle reference is the reference builder's.
//   w0..w2 = min.x, min.y, min.z of slots 0..3   w3..w5 = max.x, max.y, max.z of slots 0..3
//   w6 = bits(ref 0..3): leaf bit | triangle, or wide-node index
//   w7 = unused (keeps the stride one cache line)
// ---- triangle record: 64 B = 4 x float4, so that no record straddles two 128-byte lines (a 48-byte record does in one fetch of four)
//   t0 = { n.xyz, d }   t1 = { n1.xyz, d1 }   t2 = { n2.xyz, d2 }      the intersect part (triangle.rs:17-24), read by the leaf test
//   t3 = { n.xyz, bits(material << 2 | flags) }                         the one gather per scatter of a triangle without uv/normals
#define PTC_TRI_STRIDE 4u   /* float4 per triangle record */
// ---- triangle shading record: 80 B = 5 x float4; the common (no uv, no normals) case reads s0 only
//   s0 = { bits(material), bits(flags), uv0.x, uv0.y }
//   s1 = { uv1.x, uv1.y, uv2.x, uv2.y }
//   s2 = { vn0.x, vn0.y, vn0.z, vn1.x }   s3 = { vn1.y, vn1.z, vn2.x, vn2.y }   s4 = { vn2.z, 0, 0, 0 }

#define PTC_IMG_BILINEAR 0x80000000u   /* in a DevMaterial image reference: Texture::Bilinear instead of Nearest */
struct DevMaterial {        // 48 B
    float diffuse[3];   uint32_t diffuse_img;      // 0xFFFFFFFF = Flat, else image index | PTC_IMG_BILINEAR
    float absorb[3];    uint32_t absorb_img;
    float base_dissolve; float dissolve_value; uint32_t dissolve_img; uint32_t emit_kind;
};

struct DevImage { uint32_t width, height, channels, offset; };   // offset in floats into texel pool

struct DevSky {
    uint32_t kind; float power; float turbidity; float solar_radius;
    float zenith[3]; float sun[3];
    float configs[11][9];
    float radiances[11];
    uint32_t turb_low; float turb_frac;      // hosek_wilkie.rs:355-356, fixed per scene
};

struct DevSceneView {       // passed to kernels by value
    const float4* nodes;
    const float4* wide;         // [n_wide][8], see above
    const float4* tris;         // [n_tris][PTC_TRI_STRIDE]
    const float4* shade;
    const DevMaterial* materials;
    const DevImage* images;
    const float* texels;
    const float* refl_tab;      // [471][3]
    const float* d65_tab;       // [531]
    const float* solar_tab;     // [11][2][45][4]: the two turbidity rows the scene needs
    const float* limb_tab;      // [11][6]
    uint32_t root_ref;          // PTC_LEAF_BIT|0 for a 1-triangle scene
    uint32_t wide_depth;        // wide nodes on the longest root-to-leaf path
    uint32_t n_tris;
    uint32_t n_materials;
    uint32_t max_depth;
    uint32_t sign_select_ok;    // every box has min <= max (or is NaN) and the wide array is below 4 GiB: the EXACT kernel may
                                // load near / far planes by the sign of the ray's inverse direction (trace_step.h)
    uint32_t joins_exact;       // every child box the EXACT kernel re-derives (the two children of an odd-depth node) IS the minNum /
                                // maxNum join of that child's own two child boxes, as in every tree BVH::build or the LBVH makes.  A
                                // caller's refitted / padded tree need not be: EXACT calls on such a scene run the frame-replay kernel,
                                // which reads the stored boxes (same answer as traverse_recursive on that tree)
};

struct pt_scene {
    DevSceneView view{};
    DevSky sky{};
    DevSky* d_sky = nullptr;
    std::vector<void*> allocations;
    pt_scene_info info{};
};

// Where one traversal launch keeps its device-side scratch when that is not the context's own (grown-on-demand) buffers:
// the allocation-free entry point (trace_noalloc.hip) points these at its file-scope device array, and every slot of
// pt_trace_closest_dev_submit owns one (grow = true: spill / redo are hipMalloc'd buffers of the slot and grow like the context's).
struct ptc_trace_scratch {
    void* spill;        size_t spill_bytes;      // HBM columns of the per-lane stacks
    uint32_t* tickets;                            // PTC_TICKET_BYTES, zero before the first launch
    uint32_t* redo;     size_t redo_bytes;       // PT_TRAVERSE_EXACT: 4 bytes per ray of the batch
    uint32_t max_blocks;                          // grid cap (0: none): bounds the spill columns
    uint32_t* launch_parity;                      // host-side words that belong to `tickets`
    uint32_t* redo_parity;
    hipStream_t stream;                           // the stream the launches go to (nullptr: the context's)
    bool grow;                                    // spill / redo may be replaced by larger hipMalloc'd buffers (after `stream` has drained)
};

// One slot of pt_trace_closest_dev_submit / _wait: a stream and traversal scratch of its own, so that batches on different
// slots can overlap on the GPU.
struct ptc_trace_slot {
    bool ready = false;                                   // stream, events and ticket words exist
    hipStream_t stream = nullptr;
    ptc_trace_scratch s{};
    uint32_t launch_parity = 0, redo_parity = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;     // around everything submitted since the last wait
    uint32_t pending = 0;                                 // batches submitted since the last wait
    uint64_t rays = 0;
};

struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_in = nullptr, copy_out = nullptr;   // pt_trace_closest: uploads / downloads overlapped with the kernel
    std::vector<hipEvent_t> pipe_ev;     // pt_trace_closest: per chunk {rays uploaded, hits ready} (grown on demand)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t count_ev[8] = {};         // pt_render: one per in-flight live-count readback
    std::vector<hipEvent_t> trace_ev;    // pt_render: pairs around every traversal launch (grown on demand)
    uint32_t launch_parity = 0;          // which of the two sets of ticket words the next persistent launch uses
    uint32_t redo_parity = 0;            // which of the two redo counters the next PT_TRAVERSE_EXACT call uses
    ptc_trace_slot slots[PT_TRACE_SLOTS];   // pt_trace_closest_dev_submit: created on first use
    uint32_t traversal = PT_TRAVERSE_DEFAULT;
    std::string error;
    pt_stats stats{};
    int n_cus = 0;
    // reusable device workspaces (grown on demand, freed at shutdown)
    void* spill = nullptr;      size_t spill_bytes = 0;
    void* work = nullptr;       size_t work_bytes = 0;
    void* io = nullptr;         size_t io_bytes = 0;
    void* redo = nullptr;       size_t redo_bytes = 0;     // PT_TRAVERSE_EXACT: indices of the rays left to the replay kernel
    uint32_t* h_counters = nullptr;      // pinned
    uint32_t* ticket = nullptr;          // PTC_TICKET_BYTES on the device: two sets of ticket words (2 KiB each) of the
                                         // persistent kernels, then the two redo counters (256 B apart)
    // tuning knobs (defaults chosen by measurement; PTC_* environment variables override)
    uint32_t tune_blocks_per_cu = 6;
    uint32_t tune_refill = 16;           // idle lanes at which a wave of a persistent traversal kernel takes new rays: caller's batches
    uint32_t tune_refill_queued = 32;    // the same for the integrator's queued launches (neighbouring queue entries are neighbouring samples: larger groups share more lines)
    uint32_t tune_wide_collapse = 0;     // wide records of caller-supplied trees: 0 on the device, 1 on the host, 2 both + compare
    uint32_t tune_flatten = 0;           // caller-supplied node vectors are validated and flattened: 0 on the device, 1 on the host, 2 both + compare
    bool     tune_trust_joins = false;   // PTC_TRUST_JOINS=1 (tests only): skip the joins_exact fallback, to show what it prevents
    uint32_t tune_ticket = 64;           // rays a wave reserves per ticket atomic (after its static first block of 64)
    uint32_t tune_shade_blocks_per_cu = 256; // k_shade grid cap (blocks loop over the queue beyond it; a looping block scatters its spans over the next queue, which the traversal of that queue pays for)
    uint32_t tune_ring = 12;             // LDS ring slots per lane of the EXACT traversal kernel (8, 12 or 16)
    uint32_t tune_host_chunk = 0;          // pt_trace_closest: rays per pipelined chunk (0: by the kind of host memory, core.hip)
    uint32_t tune_host_slots = 2;          // pt_trace_closest: trace slots the chunks' kernels alternate between (1: one stream, no overlap of kernels)
    uint32_t tune_pass_paths = 128u << 20; // pt_render: paths in flight (~1.15 KiB of HBM each)
    size_t   tune_work_limit = 0;          // pt_render: workspace bytes it may reserve (0: what hipMalloc gives); beyond it the pass is halved, as when HBM is short
    uint32_t tune_sample_group = 16;       // pt_render: samples of one pixel side by side in the path ids (power of two dividing the pass's samples)
    uint32_t tune_pass_slots = 2;          // pt_render: passes in flight (1..4) of a render that does not fit in one
    float    tune_admit_frac = 0.25f;      // pt_render: the next pass joins when <= this fraction of a pass's paths is live
};

int ptc_fail(pt_ctx* ctx, int code, const std::string& msg);
int ptc_hip(pt_ctx* ctx, hipError_t e, const char* what);
int ptc_reserve(pt_ctx* ctx, void** buf, size_t* have, size_t need, const char* what);
// records ctx->trace_ev[i] on the context's stream, creating events up to i on first use
int ptc_timed_event(pt_ctx* ctx, size_t i);

#include "guard.h"

#define PTC_HIP(ctx, call)                                            \
    do {                                                              \
        int _rc = ptc_hip((ctx), (call), #call);                      \
        if (_rc != PT_OK) return _rc;                                 \
    } while (0)

#define PTC_TICKET_BYTES 8192u
#ifndef PTC_SHARDS
#define PTC_SHARDS 8u             /* ticket words / contiguous ray shards per launch of the persistent traversal kernels (power of
                                     two not required); every grid is a multiple of it (trace.hip, trace_noalloc.hip) */
#endif

// Entries per thread of the HBM stack columns.  REPLAY: one frame per binary level.  MIN_T: three pending siblings per
// wide level.  EXACT: at most a RESTORE and three pending entries per wide level (a re-pushed companion replaces two
// entries by at most two), and its redo pass replays.
inline size_t ptc_spill_depth(const DevSceneView& v, uint32_t mode) {
    const size_t d_replay = (size_t)v.max_depth + 2u, d_mint = 3u * (size_t)v.wide_depth + 2u, d_scoped = 4u * (size_t)v.wide_depth + 4u;
    return mode == PT_TRAVERSE_REPLAY ? d_replay : mode == PT_TRAVERSE_MIN_T ? d_mint : (d_scoped > d_replay ? d_scoped : d_replay);
}

// trace_noalloc.hip: resolves this device's address of the fixed scratch array (pt_init, device current)
int ptc_noalloc_setup(pt_ctx* ctx);

// launch helpers implemented in trace.hip / render.hip
int ptc_launch_trace_in(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays, const uint32_t* d_start,
                        const uint32_t* d_queue, const uint32_t* d_count, uint32_t n, pt_hit* d_out, ptc_trace_scratch* scratch);
int ptc_launch_trace(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays, const uint32_t* d_start,
                     const uint32_t* d_queue, const uint32_t* d_count, uint32_t n, pt_hit* d_out);

// lbvh.hip: device-side build into the layouts above (boxes: {min,0},{max,0} per triangle, host memory)
int ptc_build_lbvh(pt_ctx* ctx, const float4* h_boxes, uint32_t n, float4** d_nodes, float4** d_wide, uint32_t* n_wide,
                   uint32_t* max_depth, uint32_t* wide_depth, float* build_ms);

// lbvh.hip: pt_scene_create's node vector validated and flattened on the device (the host-side loop stays behind PTC_FLATTEN=1)
int ptc_flatten_nodes(pt_ctx* ctx, const pt_bvh_node* h_nodes, uint32_t n_nodes, uint32_t n_tris, float4** d_nodes,
                      uint32_t* n_internal, bool* boxes_ordered);
#define PTC_MAX_BVH_DEPTH 4096u

// the wide records of a binary node array already in HBM (same numbering and child order as the host-side collapse)
int ptc_build_wide_from_nodes(pt_ctx* ctx, const float4* d_nodes, uint32_t n_internal, float4** d_wide, uint32_t* n_wide,
                              uint32_t* max_depth, uint32_t* wide_depth, bool* joins_exact);

#endif
