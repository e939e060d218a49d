/* ptcore.h — C ABI of the MI355X render core for andreasots/pathtracer.
 *
 * The reference is one Rust crate with direct calls and no FFI; this header cuts the seam along
 * the three call sites SURVEY.md §8(b) names and declares exactly what a `ptcore-sys` crate would
 * bind (INTEGRATION.md shows that crate).  Plain C, opaque handles, caller-owned host arrays,
 * every call returns an int status (0 = ok) and never throws or aborts across the boundary: the
 * reference's panics on malformed input (scene.rs:43,53,121; bvh.rs:242-243) become error codes.
 *
 * Threading: one caller thread per pt_ctx (the reference's render job is a single rayon task
 * tree, main.rs:58).  One pt_ctx drives ONE GPU (one process per GPU; multi-GPU callers shard
 * the sample range across contexts/processes and sum the PT_RENDER_SUM buffers).
 *
 * All arithmetic is IEEE binary32, no FMA contraction, no fast-math — see DESIGN.md.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 4

/* ---- status codes ---------------------------------------------------------------------- */
enum {
    PT_OK = 0,
    PT_ERR_INVALID = 1,      /* bad argument / malformed scene (reference would panic)            */
    PT_ERR_HIP = 2,          /* HIP runtime error, text in pt_last_error()                        */
    PT_ERR_NOMEM = 3,
    PT_ERR_UNSUPPORTED = 4,  /* valid in the reference, not built yet (see DESIGN.md scope)       */
    PT_ERR_BVH_DEGENERATE = 5 /* BVH::build would hit assert!(!child_aabb.is_empty()) bvh.rs:242  */
};

#define PT_NO_TRIANGLE 0xFFFFFFFFu

/* ---- plain-data mirrors of the reference's types ---------------------------------------- */

/* AABB (bvh.rs:32-36). */
typedef struct pt_aabb { float min[3]; float max[3]; } pt_aabb;

/* enum BVHNode (bvh.rs:123-127) in `BVH.nodes` order (bvh.rs:341-343): preorder, root = [0],
 * leaves are entries of their own and hold one index into the UNPERMUTED triangle array. */
typedef struct pt_bvh_node {
    uint32_t is_leaf;        /* 1: Leaf{shape_index}; 0: Node{..}                                 */
    uint32_t shape_index;    /* Leaf                                                               */
    uint32_t child_l_index;  /* Node                                                               */
    uint32_t child_r_index;
    pt_aabb  child_l_aabb;
    pt_aabb  child_r_aabb;
} pt_bvh_node;               /* 64 bytes */

/* Arguments of Triangle::new (triangle.rs:42-49); the library recomputes n,d,n1,d1,n2,d2
 * exactly as triangle.rs:50-60 does. */
#define PT_TRI_HAS_UV      1u
#define PT_TRI_HAS_NORMALS 2u
typedef struct pt_triangle {
    float a[3], b[3], c[3];
    float uv[3][2];              /* valid iff flags & PT_TRI_HAS_UV                                */
    float vertex_normals[3][3];  /* valid iff flags & PT_TRI_HAS_NORMALS                           */
    uint32_t material_index;
    uint32_t flags;
} pt_triangle;                   /* 104 bytes */

/* Texture<P> (material.rs:80-84).  Bilinear (material.rs:101-123) is defined by the reference but never constructed by
 * its loader (Material::from_mtl builds Nearest / Flat only: material.rs:215,228,246,249); a host that builds its own
 * materials can ask for it. */
#define PT_TEX_FLAT     0u
#define PT_TEX_NEAREST  1u
#define PT_TEX_BILINEAR 2u
typedef struct pt_texture {
    uint32_t kind;
    float    value[3];   /* Flat: linear-sRGB colour, or value[0] for a scalar (dissolve) texture */
    uint32_t image;      /* Nearest / Bilinear: index into pt_scene_desc.images (< 2^31)           */
} pt_texture;

/* Image<P> (material.rs:12-16): row-major, (x, y) at texels[(x + y*width)*channels]. */
typedef struct pt_image {
    uint32_t width, height;
    uint32_t channels;   /* 3 = Color<SRGB> (already linear), 1 = f32                              */
    const float* texels;
} pt_image;

/* Material (material.rs:198-203). */
#define PT_EMIT_NULL         0u
#define PT_EMIT_ABSORBED_D65 1u
typedef struct pt_material {
    pt_texture diffuse;      /* Lambert.reflectance                                                */
    uint32_t   emit_kind;
    pt_texture absorption;   /* Emit::AbsorbedD65.absorption                                       */
    float      base_dissolve;
    pt_texture dissolve;     /* scalar                                                             */
} pt_material;

/* InitialisedSky (scene.rs:94-103) with the cooked HosekWilkieSkyModel fields
 * (hosek_wilkie.rs:36-43; both emission_correction_factor arrays are all 1.0 and omitted). */
#define PT_SKY_D65          0u
#define PT_SKY_HOSEK_WILKIE 1u
typedef struct pt_sky {
    uint32_t kind;
    float power;             /* D65                                                                */
    float configs[11][9];
    float radiances[11];
    float turbidity;
    float solar_radius;
    float zenith[3];         /* camera space, normalised (scene.rs:115-116)                        */
    float sun[3];            /* camera space, normalised (scene.rs:117-118)                        */
} pt_sky;

/* scene::Camera fields the render loop reads (scene.rs:134-140; camera.rs:11-17). */
typedef struct pt_camera {
    uint32_t width, height;
    float sensor_width, focal_length;
} pt_camera;

/* Ray::new arguments (bvh.rs:380-386); direction is used as given, NOT normalised.
 * 32 bytes so a lane fetches it as two aligned 16-byte loads. */
typedef struct pt_ray {
    float o[3]; float pad0;
    float d[3]; float pad1;
} pt_ray;

/* Option<(&Triangle, Intersection)> (bvh.rs:358-363, triangle.rs:27-33): the triangle is returned
 * by index; Intersection.normal is that triangle's `n` and is not repeated here. */
typedef struct pt_hit {
    uint32_t tri;            /* PT_NO_TRIANGLE = None                                              */
    float t, u, v;           /* Intersection.distance (ray PARAMETER), u, v                        */
} pt_hit;

typedef struct pt_scene_desc {
    const pt_triangle* triangles;  uint32_t n_triangles;
    const pt_bvh_node* nodes;      uint32_t n_nodes;      /* 0 or 2*n_triangles-1                  */
    const pt_material* materials;  uint32_t n_materials;  /* [0] = Material::default (scene.rs:204)*/
    const pt_image*    images;     uint32_t n_images;
    pt_sky sky;
} pt_scene_desc;

typedef struct pt_ctx pt_ctx;
typedef struct pt_scene pt_scene;

/* ---- lifetime --------------------------------------------------------------------------- */

/* n must be 1: one context drives one GPU.  device_ids may be NULL (device 0). */
int  pt_init(const int* device_ids, int n, pt_ctx** out);
void pt_shutdown(pt_ctx* ctx);
/* Text of the last error on this context (or of the last failed pt_init when ctx == NULL). */
const char* pt_last_error(const pt_ctx* ctx);
int  pt_abi_version(void);

/* ---- host-side mirrors of what the Rust host keeps (bvh.rs / scene.rs / material.rs) ----- */

/* BVH::build (bvh.rs:346-356 -> 134-248), bit-faithful: same split decisions, same preorder
 * node vector.  nodes_out has room for 2*n-1 entries (n >= 1); *n_nodes_out receives the count
 * (0 for n == 0).  Does not touch the GPU. */
int pt_bvh_build(const pt_triangle* triangles, uint32_t n, pt_bvh_node* nodes_out,
                 uint32_t* n_nodes_out);

/* InitialisedSky::from_sky for Sky::HosekWilkie (scene.rs:109-127) + HosekWilkieSkyModel::new
 * (hosek_wilkie.rs:46-72).  camera_transform is the 4x4 camera matrix, column-major as
 * nalgebra's Matrix4::from([[f32;4];4]) (NULL = identity).  PT_ERR_INVALID when the solar
 * elevation is not > 0 (scene.rs:121) or turbidity is outside [1, 10]. */
int pt_sky_hosek_wilkie(float turbidity, float albedo, const float zenith[3], const float sun[3],
                        const float* camera_transform, pt_sky* out);
void pt_sky_d65(float power, pt_sky* out);
/* Material::default() (material.rs:300-309). */
void pt_material_default(pt_material* out);

/* ---- before the hot path: scene ingestion ------------------------------------
 * Scene::load (scene.rs:194-282) on the host: the TOML scene file (scene.rs:64-183, serde defaults
 * included), its OBJ/MTL meshes (mesh.transform * camera_transform, quads split (a,b,c),(a,c,d),
 * Material::from_mtl material.rs:206-253) and PNG textures, then BVH::build.  Does not touch the GPU.
 * The arrays in `desc` are owned by the pt_scene_file and stay valid until pt_scene_file_free; hand
 * `desc` to pt_scene_create as it is.  Errors (the reference's Err returns and panics): PT_ERR_INVALID
 * with the text in pt_last_error(NULL). */
typedef struct pt_scene_file pt_scene_file;
typedef struct pt_scene_file_info {
    pt_scene_desc desc;
    pt_camera camera;              /* scene::Camera resolution, sensor_width, focal_length (scene.rs:134-152) */
    uint32_t samples;              /* scene::Camera.samples                                                   */
    float camera_transform[16];    /* column-major, as Matrix4::from([[f32;4];4])                             */
} pt_scene_file_info;
int  pt_scene_file_load(const char* toml_path, pt_scene_file** out);
int  pt_scene_file_get(const pt_scene_file* file, pt_scene_file_info* out);
void pt_scene_file_free(pt_scene_file* file);

/* ---- seam 1: scene upload (replaces holding Scene{bvh,triangles,materials,sky}, scene.rs:273-281)
 * Copies the caller's arrays, flattens nodes to the 64-byte device layout and triangles to the
 * 64-byte triangle record (intersect part + normal and material) + shading record, and owns the device memory until destroy. */
int  pt_scene_create(pt_ctx* ctx, const pt_scene_desc* desc, pt_scene** out);
/* Same, with the BVH built ON THE GPU instead of taken from the caller (no counterpart
 * in the reference, whose BVH::build is a host recursion): desc->nodes must be NULL.  LBVH — 63-bit Morton
 * order of the triangle boxes' centres, Karras radix tree, boxes fitted bottom-up — so the TOPOLOGY differs
 * from BVH::build: every pt_trace_closest / pt_render result is still the minimum-t hit over the same
 * triangles with the same arithmetic, only exact equal-t ties (and hits within rounding of each other, see
 * pt_set_traversal) can resolve differently, and traversal visits more nodes than in the SAH tree.
 * pt_scene_get_bvh returns the tree in the reference's own form.  pt_get_stats().total_ms afterwards = device time of the build. */
#define PT_SCENE_BUILD_LBVH 1u
int  pt_scene_create_ex(pt_ctx* ctx, const pt_scene_desc* desc, uint32_t flags, pt_scene** out);
void pt_scene_destroy(pt_ctx* ctx, pt_scene* scene);

/* ---- seam 2: closest hit (replaces self.bvh.traverse(&ray, start, &self.triangles), scene.rs:333)
 * start_tri may be NULL; an entry PT_NO_TRIANGLE means `start = None`.  Host pointers. */
int pt_trace_closest(pt_ctx* ctx, const pt_scene* scene, const pt_ray* rays,
                     const uint32_t* start_tri, uint32_t n, pt_hit* out);
/* Pinned host memory (no reference counterpart).  pt_trace_closest cuts the batch into chunks and overlaps
 * upload, kernel and download; with rays / start_tri / out in memory from pt_host_alloc every copy is an
 * asynchronous DMA and the call runs at PCIe speed, with ordinary (pageable) memory the runtime stages each
 * copy on the calling thread.  Either kind of pointer is accepted. */
int  pt_host_alloc(size_t bytes, void** out);
void pt_host_free(void* p);
/* Same with DEVICE pointers (rays/hits already resident in HBM); returns after the kernel has
 * completed.  The library works on a stream of its own: whatever the caller has queued on other streams
 * that writes d_rays / d_start_tri / d_out must be complete before the call (this holds for every
 * *_dev entry point). */
int pt_trace_closest_dev(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                         const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out);

/* The same call for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, processes whose
 * HBM is spoken for).  No counterpart in the reference.  Like pt_trace_closest_dev it runs on the library's own stream, times
 * itself with events and RETURNS AFTER THE KERNELS HAVE COMPLETED: it is a blocking call and cannot be captured into a HIP
 * graph (what it removes is the allocation, not the synchronisation).  pt_trace_closest_dev grows three device buffers on demand (stack spill
 * columns, ticket words, the redo list of PT_TRAVERSE_EXACT); this variant runs the same kernels with those buffers carved
 * out of one fixed device array that every GPU the library runs on holds until the process ends (about 144 MiB), and calls
 * no hipMalloc / hipFree.  Same results, bit for bit.  Limits, reported as PT_ERR_UNSUPPORTED with a message: at most
 * PT_NOALLOC_MAX_RAYS rays per call; a BVH so deep that fewer than 8 blocks' stack columns fit is refused (deep scenes run
 * with a smaller grid than pt_trace_closest_dev uses).  All contexts on one GPU share that GPU's array: one such call at a
 * time per GPU - calls that overlap in time on one GPU race. */
#define PT_NOALLOC_MAX_RAYS 4194304u
int pt_trace_closest_dev_noalloc(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                                 const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out);

/* pt_trace_closest_dev without the wait: the batches of consecutive samples are independent of one another (the sample loop
 * `for _ in 0..samples`, main.rs:64-76, carries nothing from one sample's rays to the next), so a caller that has the next
 * batch ready need not wait for the slowest rays of this one.  pt_trace_closest_dev_submit enqueues the batch on SLOT `slot`
 * (0 .. PT_TRACE_SLOTS-1) and returns at once; every slot has its own stream and its own traversal scratch (grown on demand
 * like the context's, so a submit may allocate).  Batches on one slot run in submission order; batches on different slots
 * may overlap on the GPU - the drain of one launch (a few long rays, most lanes idle) fills with the head of the next.
 * pt_trace_closest_dev_wait blocks until everything submitted on the slot has completed and fills pt_get_stats (trace_ms =
 * device time from the first of those kernels' start to the last one's end, trace_rays, trace_launches); waiting on an idle
 * slot returns at once.  Same kernels, same results bit for bit as pt_trace_closest_dev.  The caller keeps d_rays /
 * d_start_tri / d_out of a submitted batch alive and untouched until the wait returns; two batches in flight must not share
 * a d_out.  pt_set_traversal applies to batches submitted after it.  pt_shutdown waits for all slots. */
#define PT_TRACE_SLOTS 4u
int pt_trace_closest_dev_submit(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                                const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out, uint32_t slot);
int pt_trace_closest_dev_wait(pt_ctx* ctx, uint32_t slot);

/* Traversal flavour used by the calls above and by pt_render. All walk the caller's BVH (same boxes, same triangles,
 * same arithmetic per box and per triangle).
 *   EXACT  (default) returns what traverse_recursive returns (bvh.rs:250-338) on every ray, by construction: the same
 *          leaves are visited in the same order under the same local rule (the far child is entered iff the near child's
 *          own result is farther than the far box, bvh.rs:285; the later visit wins an equal-t tie, bvh.rs:299-303), two
 *          tree levels per memory fetch.  Rays with a non-finite component, and rays that meet a triangle at a non-finite
 *          distance, are traced by REPLAY inside the same call.  The kernel does not read the two child boxes stored in a
 *          node at odd depth: it re-derives each from that child's own two child boxes (their minNum / maxNum join, which is
 *          what BVH::build and PT_SCENE_BUILD_LBVH store).  pt_scene_create checks a caller's node vector for this; a tree
 *          whose stored boxes are looser (refitted, padded) is traversed by REPLAY, which reads the stored boxes, whenever
 *          EXACT is asked for - same answer as traverse_recursive on THAT tree, at REPLAY's rate.
 *   REPLAY the recursion frame by frame over the binary nodes; same results as EXACT, about half its rate.
 *   MIN_T  prunes against the best hit of the whole ray instead: returns a hit of minimum t, fastest; WHICH triangle of an
 *          exact equal-t tie - or of two hits an ulp or two apart where a box's entry distance rounds above the t of its own
 *          triangle - is not the recursion's choice in general (DESIGN.md 2 counts the cases a randomised sweep found). */
#define PT_TRAVERSE_MIN_T  0u
#define PT_TRAVERSE_EXACT  1u
#define PT_TRAVERSE_REPLAY 2u
#define PT_TRAVERSE_DEFAULT PT_TRAVERSE_EXACT
int pt_set_traversal(pt_ctx* ctx, uint32_t mode);

/* ---- seam 3: render (replaces the rayon closure main.rs:58-116) ---------------------------
 * Renders sample indices [spp_begin, spp_begin+spp_count) of every pixel; sample s of pixel p
 * draws from stream (seed, p, s) (pt_rng.h).  xyzw is W*H*4 floats, index 4*(x + y*W) + c
 * (main.rs:102-107).
 *   PT_RENDER_FINAL: xyz = sum * (1/spp_total), w = 1.0   (main.rs:49,100)
 *   PT_RENDER_SUM:   xyz = raw sums over the range, w = spp_count  (shard partials; add, then scale) */
#define PT_RENDER_FINAL 0u
#define PT_RENDER_SUM   1u
/* Optional flag, OR-ed into `mode`; an EXTENSION, off by default.  The reference hands evaluate_sky the bounce ray's
 * direction un-normalised (normal + unit-sphere sample, |d| up to 2: scene.rs:289-292), so under Sky::HosekWilkie
 * cos(gamma) can exceed 1 and the model's mie term becomes NaN (hosek_wilkie.rs:313-315); most pixels of a sky-lit scene
 * are then NaN, in the reference as here.  With this flag the direction is normalised (component-wise division by its
 * norm) before the sky is evaluated; everything else is unchanged. */
#define PT_RENDER_NORMALIZE_SKY_DIRECTION 0x100u
int pt_render(pt_ctx* ctx, const pt_scene* scene, const pt_camera* camera, uint32_t spp_begin,
              uint32_t spp_count, uint32_t spp_total, uint64_t seed, uint32_t mode, float* xyzw);
int pt_render_dev(pt_ctx* ctx, const pt_scene* scene, const pt_camera* camera, uint32_t spp_begin,
                  uint32_t spp_count, uint32_t spp_total, uint64_t seed, uint32_t mode,
                  float* d_xyzw);

/* ---- after the hot path: the reference's display transform ---------------
 * What renderer.rs:297-309 + shader.wgsl:32-44 do with the XYZ framebuffer: log-average luminance,
 * zones (defaults avg_zone 5, white_zone 10: renderer.rs:221-222) -> exposure + extended Reinhard on Y
 * -> XYZ to linear sRGB.  rgba receives the fragment shader's output (float, alpha passed through),
 * bgra8 what the Bgra8Unorm surface holds (clamped, quantised, NOT sRGB-encoded: renderer.rs:191);
 * either may be NULL.  stats_out (nullable) = {avg_luminance, max_luminance, middle_gray}.
 * PT_TONEMAP_SKIP_NONFINITE (no reference counterpart): NaN/inf pixels are left out of the average
 * and mapped as black; without it one NaN sample makes the whole display NaN, as in the reference. */
#define PT_TONEMAP_SKIP_NONFINITE 1u
int pt_tonemap(pt_ctx* ctx, const float* xyzw, uint32_t width, uint32_t height, float avg_zone,
               float white_zone, uint32_t flags, float* rgba, uint8_t* bgra8, float stats_out[3]);
int pt_tonemap_dev(pt_ctx* ctx, const float* d_xyzw, uint32_t width, uint32_t height, float avg_zone,
                   float white_zone, uint32_t flags, float* d_rgba, uint8_t* d_bgra8, float stats_out[3]);

/* ---- measurement ------------------------------------------------------------------------- */
typedef struct pt_stats {
    double   trace_ms;        /* HIP-event time of the traversal kernel(s) of the last call        */
    uint64_t trace_rays;      /* rays those kernels traced                                         */
    uint32_t trace_launches;
    double   total_ms;        /* HIP-event time of the whole last call (device side)               */
    uint64_t paths;           /* pt_render: camera paths started                                   */
    uint64_t segments;        /* pt_render: path segments (traversals)                             */
    /* ABI 3: pt_render's device time by kernel (HIP events around every launch on the library's stream; the rest of total_ms
     * is clears, the output kernel and gaps between launches).  Zero after calls that are not renders. */
    double   shade_ms;        /* k_shade: Material::sample + next ray (material.rs:255-297, scene.rs:333-361) */
    double   sky_ms;          /* k_sky: evaluate_sky (scene.rs:284-305)                            */
    double   accumulate_ms;   /* k_accumulate: backward fold + CMF + pixel sums (scene.rs:371-373, main.rs:93-107) */
    double   generate_ms;     /* k_generate: jitter, camera ray, hero wavelength (main.rs:77-89)  */
} pt_stats;
int pt_get_stats(const pt_ctx* ctx, pt_stats* out);

/* Scene facts the harness reports. */
typedef struct pt_scene_info {
    uint32_t n_triangles, n_internal_nodes, max_depth;
    uint64_t device_bytes;
} pt_scene_info;
int pt_scene_get_info(const pt_scene* scene, pt_scene_info* out);

/* The scene's binary tree in the form of `BVH.nodes` (bvh.rs:341-343: preorder, root = [0], leaves as entries of
 * their own) - what the reference holds after BVH::build (bvh.rs:346-356).  For a scene created from the caller's
 * nodes this returns that vector again (leaf entries with the Node fields zeroed); for PT_SCENE_BUILD_LBVH it hands
 * the device-built tree to a host that wants it as a `BVH { nodes }` of its own (the reference's traverse_recursive
 * over it then returns exactly what PT_TRAVERSE_EXACT returns here).  *n_nodes_out = 2 * n_triangles - 1 (0 for
 * an empty scene); nodes_out == NULL only queries the count; capacity is in entries. */
int pt_scene_get_bvh(pt_ctx* ctx, const pt_scene* scene, pt_bvh_node* nodes_out, uint32_t capacity,
                     uint32_t* n_nodes_out);

/* ---- harness helpers (no reference counterpart; synthetic workloads of SURVEY.md §8(d)) --- */

/* "soup" triangle generator: centre ~ U(lo, hi), two edge vectors ~ U([-s, s]^3) with
 * s = edge_scale (log_scale_hi > edge_scale: s log-uniform in [edge_scale, log_scale_hi]);
 * material 1 + (i mod n_material_cycle).  Deterministic in (seed, i) via pt_rng.h. */
int pt_gen_soup(uint64_t seed, uint32_t n, const float lo[3], const float hi[3], float edge_scale,
                float log_scale_hi, uint32_t n_material_cycle, pt_triangle* out);
/* Pixel-centre camera rays, Camera::generate_ray(x, y) (camera.rs:19-28), row-major. */
int pt_gen_primary_rays(const pt_camera* camera, pt_ray* out);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
