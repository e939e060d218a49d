// render.hip — wavefront spectral path tracer: generate -> [extend -> shade -> compact]* -> accumulate.
//
// Replaces the rayon sample loop and Scene::radiance:
//   main.rs:64-109     per pixel, per sample: jitter, camera ray, hero wavelengths, radiance, XYZ
//   scene.rs:307-374   the path loop (<= 16 scatter vertices, <= 1024 iterations, Russian roulette
//                      from the 3rd vertex, dissolve pass-through, sky on miss, back-to-front fold)
//   material.rs:255-297  Material::sample (dissolve test, normal, Lambert sphere offset, emission)
//
// One PASS renders `spp_pass` consecutive sample indices of every pixel at once.  Up to PTC_MAX_SLOTS passes are in
// flight together, each in its own SLOT of the workspace: path id p = slot * cap + position in the pass, where the samples
// of a pixel sit side by side in groups of up to 16 (path_to_pixel_sample below), so that a traversal wave holds a few
// pixels' worth of samples: rays that start together.  When the
// live paths of the passes in flight have thinned out, the next pass is generated into a free slot and its paths are
// appended to the same queue, so that a traversal launch never holds only the thin tail of one pass (the last ten of
// a pass's ~18 iterations hold < 3 % of its paths and cost a ray's latency each).  A pass is folded into the image
// when the host learns that its last path has ended, oldest pass first, so every pixel still receives its samples in
// sample order.  The state of a LIVE path (ray, rng index, state word, hit) lives at a POSITION, not at the path id: every
// k_shade writes the survivors of a span of the queue side by side into the other of two buffers, so that after any number
// of bounces the live paths are dense in memory (PassState below); the QUEUES hold positions.  Only what a path leaves
// behind - its vertices, and 32 bytes when it ends - is indexed by the path id; every access is a whole 16/32/64-byte
// record.  Each iteration the traversal kernel (trace.hip) extends every queued ray, then the shade kernel consumes the
// hits, writes one {weight, emit} vertex per scatter into the path's vertex column and appends the surviving paths to the
// next queue (ballot / popcount ranks into a per-block LDS list, one global atomic per list and 2 048 paths).
//
// The vertex column stores the sub-path exactly as the reference's ArrayVec<Bounce, 16> does, so
// the accumulate kernel can fold it back to front, L = L*w + e (scene.rs:371-373), in the
// reference's order — with 288 GB of HBM the 1 KiB per path is cheap, and a bounce costs one 64-B
// write per path instead of a read-modify-write of throughput and radiance.  Each pixel's samples
// are added in sample order by the one lane that owns the pixel, so the image is a deterministic
// function of (scene, camera, seed, sample range) and equals the CPU checker's bit for bit
// wherever the two agree on every hit.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_types.h"
#include "shading.h"
#include "../../include/pt_rng.h"

#define PTC_BLOCK 256
#define PTC_SHADE_SPAN 2048u     /* queue entries a k_shade block handles per global queue reservation */
#define PTC_MAX_VERTS 16        /* ArrayVec<_, 16>, scene.rs:324 */
#define PTC_MAX_ITERS 1024u     /* scene.rs:328 */
#define PTC_MAX_SLOTS 4          /* passes in flight */

namespace {

struct PassState {
    // LIVE paths are addressed by POSITION, not by path id: every iteration's k_shade writes the ray and state of each
    // survivor into the other of two buffers, at (first queue index of its span) + (its rank among the span's survivors),
    // so the survivors of a span are neighbours in memory and the next iteration's traversal and shading read runs of
    // whole cache lines however few of the pass's paths are still alive (by path id, every array cost one 128-byte
    // request per path and kernel from the third iteration on).  The queues hold positions.
    float4*   pray[2];    // [CAP][2]  {origin.xyz, bits(start triangle)}, {direction.xyz, hero wavelength offset}:
                          //           pt_ray layout with the two pads in use; trace.hip reads the start from pad0
    uint4*    pmeta[2];   // [CAP]     {path id, draws consumed so far by the path's stream, len (5 bits) | iterations << 5
                          //            (11 bits) | emit mask << 16 (bit k set = vertex k carries an emission term), -}
    uint4*    hits;       // [CAP]     by position, written by the traversal of the iteration
    float4*   tail;       // [N][2]    by path id, written when the path ENDS: {last direction.xyz, hero}, {bits(state word), -, -, -}
                          //           - what k_sky and k_accumulate need
    float4*   verts;      // weights [16][N][2] then emits [16][N][2]: vertex k of path p = {weight[0..3], weight[4..7]} and
                          //         {emit[0..3], emit[4..7]}, vertex-major so that the lanes of an iteration (k_shade) and the
                          //         neighbouring pixels of a sample (k_accumulate) touch neighbouring 32-byte records; the emit
                          //         record is only written and read where the state's emit mask says so (7 of 8 soup materials
                          //         emit nothing)
    uint32_t* queue[2];   // [N] each: positions of the live paths
    uint32_t* miss_queue; // [N] paths whose ray left the scene since the last k_sky
    uint32_t* counts;     // [8]: three rotating pairs {live paths, live paths of the watched slot} at [2c], [2c + 1]
                          //      (iteration i reads pair i%3, fills pair (i+1)%3, clears pair (i+2)%3); [6] = length of miss_queue
    uint32_t  cap;        // paths per slot
    uint32_t  N;          // slots * cap: the path id space
    uint32_t  npix;
    uint32_t  sample_base[PTC_MAX_SLOTS];   // first sample index of the pass in each slot
    uint32_t  group[PTC_MAX_SLOTS];         // samples of one pixel that sit side by side in the path ids of the slot's pass
};

// Path j of a pass <-> (pixel, sample of the pass): the samples of a pixel come in groups of G consecutive path ids, group
// after group of all pixels: j = (s / G) * (npix * G) + pixel * G + s % G.  G = 1 is sample-major (a wave = 64 neighbouring
// pixels of one sample); G = 16 makes a wave 4 pixels x 16 samples.
__device__ __forceinline__ void path_to_pixel_sample(const PassState& st, uint32_t slot, uint32_t j, uint32_t& pixel, uint32_t& s) {
    const uint32_t G = st.group[slot], span = st.npix * G;
    const uint32_t blk = j / span, r = j - blk * span;
    pixel = r / G;
    s = blk * G + (r - pixel * G);
}

__device__ __forceinline__ uint32_t slot_of(const PassState& st, uint32_t p) {
    return (uint32_t)(p >= st.cap) + (uint32_t)(p >= 2u * st.cap) + (uint32_t)(p >= 3u * st.cap);   // cap <= 2^28
}
__device__ __forceinline__ float4* weight_ptr(const PassState& st, uint32_t p, uint32_t k) {
    return st.verts + ((size_t)k * st.N + p) * 2u;
}
__device__ __forceinline__ float4* emit_ptr(const PassState& st, uint32_t p, uint32_t k) {
    return st.verts + ((size_t)(PTC_MAX_VERTS + k) * st.N + p) * 2u;
}
__device__ __forceinline__ void store_vertex(const PassState& st, uint32_t p, uint32_t k, const float w[PTC_NW],
                                             const float e[PTC_NW], bool has_emit) {
    float4* v = weight_ptr(st, p, k);
    v[0] = make_float4(w[0], w[1], w[2], w[3]);
    v[1] = make_float4(w[4], w[5], w[6], w[7]);
    if (has_emit) {
        float4* ev = emit_ptr(st, p, k);
        ev[0] = make_float4(e[0], e[1], e[2], e[3]);
        ev[1] = make_float4(e[4], e[5], e[6], e[7]);
    }
}
#define PTC_STATE_LEN(y)   ((y) & 0x1Fu)
#define PTC_STATE_ITER(y)  (((y) >> 5) & 0x7FFu)
#define PTC_STATE_EMASK(y) ((y) >> 16)
#define PTC_STATE_PACK(len, iter, emask) ((len) | ((iter) << 5) | ((emask) << 16))

// ---- generate: main.rs:76-89 -----------------------------------------------------------------------
// The `n` paths of a new pass go into `slot` and are appended to the queue of the iteration about to be enqueued, behind
// the survivors already there (counter pair `c`; k_admit then adds n to it - a kernel of its own because every block of
// this one reads the old length).
__global__ __launch_bounds__(PTC_BLOCK) void k_generate(PassState st, pt_camera cam, uint64_t seed, uint32_t slot,
                                                       uint32_t n, uint32_t c, uint32_t qsel) {
    const uint32_t j = blockIdx.x * PTC_BLOCK + threadIdx.x;
    if (j >= n) return;
    const uint32_t p = slot * st.cap + j;
    uint32_t pixel, s_local;
    path_to_pixel_sample(st, slot, j, pixel, s_local);
    const uint32_t sample = st.sample_base[slot] + s_local;
    const uint32_t px = pixel % cam.width, py = pixel / cam.width;
    pt_rng rng;
    pt_rng_init(&rng, seed, pixel, sample, 0u);
    const float x = (float)px + pt_rng_normal(&rng) * 0.5f;                     // main.rs:77
    const float y = (float)py + pt_rng_normal(&rng) * 0.5f;                     // main.rs:78
    // Camera::new + generate_ray (camera.rs:11-28)
    const float w = (float)cam.width, h = (float)cam.height;
    const float sx = 1.0f / (w + 1.0f), sy = 1.0f / (h + 1.0f);
    const float sensor1 = cam.sensor_width * h / w;
    const float dx = ((x + 0.5f) * sx - 0.5f) * cam.sensor_width;
    const float dy = -((y + 0.5f) * sy - 0.5f) * sensor1;
    const float dz = -cam.focal_length;
    const float hero = pt_rng_uniform(&rng, 0.0f, pt_uniform_scale(0.0f, (830.0f - 360.0f) / (float)PTC_NW));  // main.rs:83-84
    // positions: behind whatever the last iteration's k_shade may have written into this buffer (it wrote below the length
    // of ITS queue, counter pair c - 1, rounded up to a span)
    const uint32_t prev = st.counts[2u * ((c + 2u) % 3u)];
    const uint32_t pos = (prev + PTC_SHADE_SPAN - 1u) / PTC_SHADE_SPAN * PTC_SHADE_SPAN + j;
    st.pray[qsel][2u * (size_t)pos] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(PT_NO_TRIANGLE));
    st.pray[qsel][2u * (size_t)pos + 1u] = make_float4(dx, dy, dz, hero);
    st.pmeta[qsel][pos] = make_uint4(p, rng.index, 0u, 0u);
    st.queue[qsel][st.counts[2u * c] + j] = pos;
}
__global__ void k_admit(uint32_t* counts, uint32_t c, uint32_t n, uint32_t watched) {
    if (threadIdx.x == 0u && blockIdx.x == 0u) {
        counts[2u * c] += n;
        if (watched) counts[2u * c + 1u] += n;       // the new pass is the oldest in flight
    }
}

// ---- shade: one iteration of the loop body scene.rs:333-368 ------------------------------------------
// The live count is read from the device (st.counts[cin]) so that the host never has to wait for it:
// the grid is sized for an upper bound and surplus threads only take part in the ballots.
//
// Everything that depends only on the hit (ray, state, shading record, triangle normal) is requested in
// one batch, the spectral tables live in LDS, and the paths that go on (their new positions to the next
// iteration's queue, the ids of those that missed to the miss queue) are collected per block in LDS and
// appended with ONE global atomic per list and span of PTC_SHADE_SPAN paths.
__global__ __launch_bounds__(PTC_BLOCK) void k_shade(DevSceneView sc, PassState st, uint32_t cur, uint32_t cin,
                                                    uint64_t seed, uint32_t watch_slot) {
    __shared__ ptc::SpectralLds tab;
    // what this block sends on: the survivors' new POSITIONS fill qbuf from the front, the ids of the paths that missed
    // from the back (they are disjoint subsets of the PTC_SHADE_SPAN paths a block handles per span)
    __shared__ uint32_t qbuf[PTC_SHADE_SPAN];
    __shared__ uint32_t q_n[3], q_base[2];              // continuing, missed, continuing in the watched slot
    const uint32_t count = st.counts[2u * cin];
    uint32_t* const next_count = &st.counts[2u * (cin == 2u ? 0u : cin + 1u)];
    if (blockIdx.x == 0u && threadIdx.x < 2u) st.counts[2u * (cin == 0u ? 2u : cin - 1u) + threadIdx.x] = 0u;   // the pair the NEXT iteration fills
    if (blockIdx.x * PTC_SHADE_SPAN >= count) return;            // surplus block of an over-sized grid
    ptc::spectral_lds_fill(tab, sc.refl_tab, sc.d65_tab, threadIdx.x, PTC_BLOCK);
    const uint32_t lane = threadIdx.x & 63u;
    // grid-stride over the queue in spans of PTC_SHADE_SPAN entries (trip counts are uniform per block)
    for (uint32_t span0 = blockIdx.x * PTC_SHADE_SPAN; span0 < count; span0 += gridDim.x * PTC_SHADE_SPAN) {
    if (threadIdx.x < 3u) q_n[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t k0 = span0; k0 < span0 + PTC_SHADE_SPAN && k0 < count; k0 += PTC_BLOCK) {
        const uint32_t k = k0 + threadIdx.x;
        const bool active = k < count;
        const uint32_t pos = active ? st.queue[cur][k] : 0u;
        uint4 hit = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u), meta = make_uint4(0u, 0u, 0u, 0u);
        // new_d starts as the path's direction and new_state as its state: what a path that passes straight through, or
        // ends at the sky, carries on (and no second copy of them is held in registers across the body)
        float4 new_d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (active) { hit = st.hits[pos]; meta = st.pmeta[cur][pos]; new_d = st.pray[cur][2u * (size_t)pos + 1u]; }
        const uint32_t p = meta.x;
        // miss: the path ends with {0, sky} (scene.rs:362-367); the sky term is evaluated by k_sky, once
        // per pass over all such paths, so that this kernel's lanes do not wait for it
        const bool missed = active && hit.x == PT_NO_TRIANGLE;
        const bool shading = active && !missed;

        bool cont = false, in_watched = false;
        float4 new_o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint2 new_state = make_uint2(meta.y, meta.z);

        if (shading) {
            const uint32_t tri = hit.x;
            // one batch of loads: all of them depend on (pos, tri) only
            const float4 ro = st.pray[cur][2u * (size_t)pos], rd = new_d;
            const uint2 state = make_uint2(meta.y, meta.z);
            const float4* srec = sc.shade + (size_t)tri * 5u;
            const float4 t0 = sc.tris[(size_t)tri * PTC_TRI_STRIDE + 3u];                                     // Intersection.normal = tri.n, material | flags
            uint32_t len = PTC_STATE_LEN(state.y), iter = PTC_STATE_ITER(state.y), emask = PTC_STATE_EMASK(state.y);
            float wl[PTC_NW];
            ptc::wavelengths(rd.w, wl);
            const float dist = __uint_as_float(hit.y), hu = __uint_as_float(hit.z), hv = __uint_as_float(hit.w);
            const uint32_t mat_id = __float_as_uint(t0.w) >> 2, flags = __float_as_uint(t0.w) & 3u;
            const DevMaterial m = sc.materials[mat_id];
            float u = hu, v = hv;
            if (flags & PT_TRI_HAS_UV) {                                        // material.rs:266-270
                const float4 s0 = srec[0], s1 = srec[1];
                const float w0 = 1.0f - hu - hv;
                u = w0 * s0.z + hu * s1.x + hv * s1.z;
                v = w0 * s0.w + hu * s1.y + hv * s1.w;
            }
            pt_rng rng;
            const uint32_t slot = slot_of(st, p), j = p - slot * st.cap;
            in_watched = slot == watch_slot;
            uint32_t pixel, s_local;
            path_to_pixel_sample(st, slot, j, pixel, s_local);
            pt_rng_init(&rng, seed, pixel, st.sample_base[slot] + s_local, state.x);
            float dis = m.dissolve_value;
            if (m.dissolve_img != 0xFFFFFFFFu) {
                float g, b; const float flat[3] = {dis, dis, dis};
                ptc::texture_rgb(sc, m.dissolve_img, flat, u, v, dis, g, b);
            }
            const float px = ro.x + rd.x * dist, py = ro.y + rd.y * dist, pz = ro.z + rd.z * dist;   // material.rs:292
            new_o = make_float4(px, py, pz, __uint_as_float(tri));              // start = Some(triangle), scene.rs:356,360
            if (pt_rng_f32(&rng) >= m.base_dissolve * dis) {                    // material.rs:275
                cont = true;                                                    // dissolved: pass straight through (scene.rs:357-361)
            } else {
                float nx = t0.x, ny = t0.y, nz = t0.z;
                if (flags & PT_TRI_HAS_NORMALS) {                               // material.rs:279-282
                    const float4 s2 = srec[2], s3 = srec[3], s4 = srec[4];
                    const float w0 = 1.0f - hu - hv;
                    nx = w0 * s2.x + hu * s2.w + hv * s3.z;
                    ny = w0 * s2.y + hu * s3.x + hv * s3.w;
                    nz = w0 * s2.z + hu * s3.y + hv * s4.x;
                }
                if (!(ptc::dot3(nx, ny, nz, rd.x, rd.y, rd.z) < 0.0f)) { nx = -nx; ny = -ny; nz = -nz; }   // :288
                const float nlen = pt_sqrt(ptc::dot3(nx, ny, nz, nx, ny, nz));  // :290 normalize()
                nx = nx / nlen; ny = ny / nlen; nz = nz / nlen;
                float sph[3];
                pt_rng_unit_sphere(&rng, sph);                                  // material.rs:168
                new_d = make_float4(nx + sph[0], ny + sph[1], nz + sph[2], rd.w);
                // the vertex is stored HERE, not after the queue hand-off below: its 16 registers are dead before the ballots
                bool has_emit = false;
                float weight[PTC_NW], emit[PTC_NW];
#pragma unroll
                for (int i = 0; i < PTC_NW; ++i) emit[i] = 0.0f;
                float cr, cg, cb;
                ptc::texture_rgb(sc, m.diffuse_img, m.diffuse, u, v, cr, cg, cb);
                // The empty asm makes wavelength i+1 "depend" on the result of wavelength i-1: without it the
                // compiler puts the table rows of all eight wavelengths in flight at once (64+ VGPRs) and the
                // kernel drops to two waves per SIMD.  Two wavelengths in flight cover the LDS latency.
#ifdef PTC_EMU      /* the CPU build of this file (tests/cpp/emu_rt.h) has no "v" registers and no need */
#define PTC_AFTER(next, done) do { } while (0)
#else
#define PTC_AFTER(next, done) asm volatile("" : "+v"(next) : "v"(done))
#endif
#pragma unroll
                for (int i = 0; i < PTC_NW; ++i) {
                    weight[i] = ptc::reflectance_at_one(tab, cr, cg, cb, wl[i]);
                    if (i >= 1 && i + 1 < PTC_NW) PTC_AFTER(wl[i + 1], weight[i - 1]);
                }
                if (m.emit_kind == PT_EMIT_ABSORBED_D65) {                      // material.rs:183-185
                    has_emit = true;
                    emask |= 1u << len;
                    float ar, ag, ab;
                    ptc::texture_rgb(sc, m.absorb_img, m.absorb, u, v, ar, ag, ab);
#pragma unroll
                    for (int i = 0; i < PTC_NW; ++i) {
                        emit[i] = ptc::d65_one(tab, wl[i]) * ptc::reflectance_at_one(tab, ar, ag, ab, wl[i]);
                        if (i >= 1 && i + 1 < PTC_NW) PTC_AFTER(wl[i + 1], emit[i - 1]);
                    }
                }
                bool survive = true;
                if (len > 1u) {                                                 // scene.rs:342-353
                    float mx = weight[0];
#pragma unroll
                    for (int i = 1; i < PTC_NW; ++i) mx = (weight[i] > mx) ? weight[i] : mx;
                    if (pt_rng_f32(&rng) < mx) {
#pragma unroll
                        for (int i = 0; i < PTC_NW; ++i) weight[i] = weight[i] / mx;
                    } else {
                        survive = false;
#pragma unroll
                        for (int i = 0; i < PTC_NW; ++i) weight[i] = 0.0f;
                    }
                }
                store_vertex(st, p, len, weight, emit, has_emit);
                len += 1u;
                cont = survive;                                                 // scene.rs:354-356 (new_d is the new direction)
            }
            iter += 1u;
            // loop header: `for _ in 0..1024` and the capacity check scene.rs:328-331
            cont = cont && (len < PTC_MAX_VERTS) && (iter < PTC_MAX_ITERS);
            new_state = make_uint2(rng.index, PTC_STATE_PACK(len, iter, emask));
        }

        // hand the path on: ballot -> one LDS atomic per wave and list -> slot in the block's buffer
        const unsigned long long mmask = __ballot(missed), cmask = __ballot(cont), wmask = __ballot(cont && in_watched);
        const unsigned long long below = (1ull << lane) - 1ull;
        uint32_t new_pos = 0u;
        if (cmask != 0ull) {
            const int leader = __ffsll((long long)cmask) - 1;
            uint32_t base = 0u;
            if ((int)lane == leader) {
                base = atomicAdd(&q_n[0], (uint32_t)__popcll(cmask));
                if (wmask != 0ull) atomicAdd(&q_n[2], (uint32_t)__popcll(wmask));
            }
            base = __shfl(base, leader);
            // the survivor's position in the other buffer: its rank among the span's survivors, from the span's first index
            if (cont) { new_pos = span0 + base + (uint32_t)__popcll(cmask & below); qbuf[new_pos - span0] = new_pos; }
        }
        if (mmask != 0ull) {
            const int leader = __ffsll((long long)mmask) - 1;
            uint32_t base = 0u;
            if ((int)lane == leader) base = atomicAdd(&q_n[1], (uint32_t)__popcll(mmask));
            base = __shfl(base, leader);
            if (missed) qbuf[PTC_SHADE_SPAN - 1u - (base + (uint32_t)__popcll(mmask & below))] = p;
        }

        if (cont) {
            st.pray[cur ^ 1u][2u * (size_t)new_pos] = new_o;
            st.pray[cur ^ 1u][2u * (size_t)new_pos + 1u] = new_d;
            st.pmeta[cur ^ 1u][new_pos] = make_uint4(p, new_state.x, new_state.y, 0u);
        } else if (active) {              // the path ends: here (k_accumulate wants its hero wavelength, new_d.w) or at the sky
            st.tail[2u * (size_t)p] = new_d;
            st.tail[2u * (size_t)p + 1u] = make_float4(__uint_as_float(new_state.y), 0.0f, 0.0f, 0.0f);
        }
    }
    // one global atomic per list and SPAN (a single counter word takes ~90 returning atomics per
    // microsecond chip-wide: one per wave, as this kernel first did, made those atomics its whole run time)
    __syncthreads();
    if (threadIdx.x == 0u && q_n[0] != 0u) q_base[0] = atomicAdd(next_count, q_n[0]);
    if (threadIdx.x == 64u && q_n[1] != 0u) q_base[1] = atomicAdd(&st.counts[6], q_n[1]);
    if (threadIdx.x == 128u && q_n[2] != 0u) atomicAdd(next_count + 1, q_n[2]);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < q_n[0]; i += PTC_BLOCK) st.queue[cur ^ 1u][q_base[0] + i] = qbuf[i];
    for (uint32_t i = threadIdx.x; i < q_n[1]; i += PTC_BLOCK) st.miss_queue[q_base[1] + i] = qbuf[PTC_SHADE_SPAN - 1u - i];
    __syncthreads();
    }   // spans
}

// ---- sky: Scene::evaluate_sky for every path that left the scene since the last k_sky (scene.rs:362-367); the kernel
// that follows (k_accumulate) empties the list ------
__global__ __launch_bounds__(PTC_BLOCK) void k_sky(DevSceneView sc, const DevSky* __restrict__ sky, PassState st, uint32_t normalize) {
    const uint32_t count = st.counts[6];
    for (uint32_t k = blockIdx.x * PTC_BLOCK + threadIdx.x; k < count; k += gridDim.x * PTC_BLOCK) {
        const uint32_t p = st.miss_queue[k];
        const float4 rd = st.tail[2u * (size_t)p];
        const uint32_t sy = __float_as_uint(st.tail[2u * (size_t)p + 1u].x);
        const uint32_t len = PTC_STATE_LEN(sy);
        float wl[PTC_NW], e[PTC_NW], zero[PTC_NW];
        ptc::wavelengths(rd.w, wl);
        float dx = rd.x, dy = rd.y, dz = rd.z;
        if (normalize) {                           // PT_RENDER_NORMALIZE_SKY_DIRECTION (extension, ptcore.h)
            const float n = pt_sqrt(ptc::dot3(dx, dy, dz, dx, dy, dz));
            dx = dx / n; dy = dy / n; dz = dz / n;
        }
        ptc::evaluate_sky(sc, sky, dx, dy, dz, wl, e);
#pragma unroll
        for (int i = 0; i < PTC_NW; ++i) zero[i] = 0.0f;
        store_vertex(st, p, len, zero, e, true);
        st.tail[2u * (size_t)p + 1u].x = __uint_as_float((sy + 1u) | (1u << (16u + len)));     // len + 1, emit bit of the sky vertex
    }
}

// ---- accumulate: fold + CMF + pixel sum (scene.rs:371-373, main.rs:93-97) -----------------------------
// A block takes 256 consecutive path ids of one sample group = 256 / G pixels x G samples (coalesced reads of the path state and
// of the vertex records), every thread folds ITS path back to front and weights it with the CMF, the 24 terms go through LDS,
// and the first thread of each pixel adds them to the pixel in sample order, wavelength by wavelength - the additions and
// their order are those of main.rs:93-97 whatever G is.  The block then moves on to the same pixels' next sample group.
__global__ __launch_bounds__(PTC_BLOCK) void k_accumulate(PassState st, uint32_t slot, uint32_t spp_pass, float4* __restrict__ acc) {
    __shared__ float terms[PTC_BLOCK][3 * PTC_NW + 1];        // + 1: threads of a wave start in different banks
    if (blockIdx.x == 0u && threadIdx.x == 0u) st.counts[6] = 0u;      // the k_sky before this kernel has consumed the miss list
    const uint32_t G = st.group[slot];                        // a power of two <= 64 that divides spp_pass
    const uint32_t per_block = PTC_BLOCK / G;
    const uint32_t local = threadIdx.x / G, g = threadIdx.x - local * G;
    const uint32_t pixel = blockIdx.x * per_block + local;
    const bool live = pixel < st.npix, owner = live && g == 0u;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (owner) a = acc[pixel];
    for (uint32_t blk = 0; blk * G < spp_pass; ++blk) {
        if (live) {
            const uint32_t p = slot * st.cap + blk * (st.npix * G) + pixel * G + g;
            const uint32_t sy = __float_as_uint(st.tail[2u * (size_t)p + 1u].x);
            const uint32_t len = PTC_STATE_LEN(sy), emask = PTC_STATE_EMASK(sy);
            const float hero = st.tail[2u * (size_t)p].w;
            float L[PTC_NW];
#pragma unroll
            for (int i = 0; i < PTC_NW; ++i) L[i] = 0.0f;
            for (uint32_t k = len; k-- > 0u;) {
                const float4* v = weight_ptr(st, p, k);
                const float4 w0 = v[0], w1 = v[1];
                float4 e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), e1 = e0;      // Emit::Null samples to 0 (material.rs:187)
                if ((emask >> k) & 1u) { const float4* ev = emit_ptr(st, p, k); e0 = ev[0]; e1 = ev[1]; }
                L[0] = L[0] * w0.x + e0.x; L[1] = L[1] * w0.y + e0.y; L[2] = L[2] * w0.z + e0.z; L[3] = L[3] * w0.w + e0.w;
                L[4] = L[4] * w1.x + e1.x; L[5] = L[5] * w1.y + e1.y; L[6] = L[6] * w1.z + e1.z; L[7] = L[7] * w1.w + e1.w;
            }
            float wl[PTC_NW];
            ptc::wavelengths(hero, wl);
#pragma unroll
            for (int i = 0; i < PTC_NW; ++i) {
                float X, Y, Z;
                ptc::from_wavelength(wl[i], X, Y, Z);
                terms[threadIdx.x][3 * i + 0] = X * L[i] * (1.0f / (float)PTC_NW);
                terms[threadIdx.x][3 * i + 1] = Y * L[i] * (1.0f / (float)PTC_NW);
                terms[threadIdx.x][3 * i + 2] = Z * L[i] * (1.0f / (float)PTC_NW);
            }
        }
        __syncthreads();
        if (owner) {
            for (uint32_t s = 0; s < G; ++s) {
                const float* t = terms[threadIdx.x + s];
#pragma unroll
                for (int i = 0; i < PTC_NW; ++i) { a.x += t[3 * i + 0]; a.y += t[3 * i + 1]; a.z += t[3 * i + 2]; }
            }
        }
        __syncthreads();
    }
    if (owner) acc[pixel] = a;
}

// main.rs:100-107 (FINAL) or raw sums (SUM)
__global__ __launch_bounds__(PTC_BLOCK) void k_output(const float4* __restrict__ acc, float4* __restrict__ out,
                                                     uint32_t npix, uint32_t mode, float inv_total, float count) {
    const uint32_t pixel = blockIdx.x * PTC_BLOCK + threadIdx.x;
    if (pixel >= npix) return;
    const float4 a = acc[pixel];
    if (mode == PT_RENDER_FINAL) out[pixel] = make_float4(a.x * inv_total, a.y * inv_total, a.z * inv_total, 1.0f);
    else out[pixel] = make_float4(a.x, a.y, a.z, count);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

static int render_impl(pt_ctx* ctx, const pt_scene* scene, const pt_camera* cam, uint32_t spp_begin,
                       uint32_t spp_count, uint32_t spp_total, uint64_t seed, uint32_t mode, float* d_xyzw);

static int pt_render_dev_body(pt_ctx* ctx, const pt_scene* scene, const pt_camera* cam, uint32_t spp_begin,
                             uint32_t spp_count, uint32_t spp_total, uint64_t seed, uint32_t mode, float* d_xyzw) {
    if (!ctx) return PT_ERR_INVALID;
    const int rc = render_impl(ctx, scene, cam, spp_begin, spp_count, spp_total, seed, mode, d_xyzw);
    // an error can leave kernels of the aborted render queued: drain them so that the workspace and the
    // events are quiet before the caller's next call (the error text is kept)
    if (rc != PT_OK && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

extern "C" int pt_render_dev(pt_ctx* ctx, const pt_scene* scene, const pt_camera* cam, uint32_t spp_begin,
                             uint32_t spp_count, uint32_t spp_total, uint64_t seed, uint32_t mode, float* d_xyzw) {
    return ptc_guard(ctx, "pt_render_dev", [&] { return pt_render_dev_body(ctx, scene, cam, spp_begin, spp_count, spp_total, seed, mode, d_xyzw); });
}

static int render_impl(pt_ctx* ctx, const pt_scene* scene, const pt_camera* cam, uint32_t spp_begin,
                       uint32_t spp_count, uint32_t spp_total, uint64_t seed, uint32_t mode, float* d_xyzw) {
    if (!scene || !cam || !d_xyzw) return ptc_fail(ctx, PT_ERR_INVALID, "pt_render_dev: NULL argument");
    if (cam->width == 0 || cam->height == 0) return ptc_fail(ctx, PT_ERR_INVALID, "pt_render_dev: empty image");
    const uint32_t normalize_sky = mode & PT_RENDER_NORMALIZE_SKY_DIRECTION;
    mode &= ~PT_RENDER_NORMALIZE_SKY_DIRECTION;
    if (mode != PT_RENDER_FINAL && mode != PT_RENDER_SUM) return ptc_fail(ctx, PT_ERR_INVALID, "pt_render_dev: unknown mode");
    if (mode == PT_RENDER_FINAL && spp_total == 0) return ptc_fail(ctx, PT_ERR_INVALID, "pt_render_dev: spp_total is 0");
    const uint64_t npix64 = (uint64_t)cam->width * cam->height;
    if (npix64 > (1ull << 28)) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_render_dev: image larger than 2^28 pixels");
    if ((uint64_t)spp_begin + spp_count > 0xFFFFFFFFull) return ptc_fail(ctx, PT_ERR_INVALID, "pt_render_dev: sample range overflows");
    const uint32_t npix = (uint32_t)npix64;
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    ctx->stats = pt_stats{};

    // ctx->tune_pass_paths paths in flight (1.15 KiB of HBM each).  A render that fits is ONE pass; a longer one is cut
    // into passes of whole samples, ctx->tune_pass_slots of them in flight, each 1 / slots of the budget.  When the
    // workspace does not fit in what is left of HBM the pass is halved until it does (one sample of every pixel is the
    // floor, then fewer slots): that costs throughput, never the image (samples are independent and each pixel adds its
    // own in sample order).
    uint32_t slots = 1u;
    uint32_t spp_pass = (uint32_t)std::max<uint64_t>(1, (uint64_t)ctx->tune_pass_paths / npix);
    if (spp_pass >= spp_count) spp_pass = spp_count ? spp_count : 1u;
    else {
        slots = std::min<uint32_t>(ctx->tune_pass_slots, PTC_MAX_SLOTS);
        spp_pass = std::max(1u, spp_pass / slots);
    }
    size_t o_acc, o_ray0, o_ray1, o_meta0, o_meta1, o_hits, o_tail, o_q0, o_q1, o_qm, o_cnt, o_verts;
    for (;;) {
        const uint32_t n_pass = spp_count ? (spp_count + spp_pass - 1u) / spp_pass : 1u;
        slots = std::max(1u, std::min(slots, n_pass));
        const size_t N = (size_t)slots * spp_pass * npix;
        // ---- carve the workspace -----------------------------------------------------------------
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
        o_acc = carve((size_t)npix * sizeof(float4));
        const size_t CAP = N + PTC_SHADE_SPAN;            // positions: a new pass starts on a span boundary behind the live paths
        o_ray0 = carve(CAP * 2 * sizeof(float4)); o_ray1 = carve(CAP * 2 * sizeof(float4));
        o_meta0 = carve(CAP * sizeof(uint4)); o_meta1 = carve(CAP * sizeof(uint4));
        o_hits = carve(CAP * sizeof(uint4));
        o_tail = carve(N * 2 * sizeof(float4));
        o_q0 = carve(N * 4); o_q1 = carve(N * 4); o_qm = carve(N * 4);
        o_cnt = carve(256);
        o_verts = carve(N * PTC_MAX_VERTS * 2 * PTC_NW * sizeof(float));
        int rc = PT_ERR_NOMEM;
        if (N <= (1ull << 30) && (ctx->tune_work_limit == 0u || off <= ctx->tune_work_limit))
            rc = ptc_reserve(ctx, &ctx->work, &ctx->work_bytes, off, "render workspace");
        if (rc == PT_OK) break;
        if (rc != PT_ERR_NOMEM || (spp_pass == 1u && slots <= 1u)) return rc == PT_ERR_NOMEM && ctx->error.empty() ? ptc_fail(ctx, rc, "pt_render_dev: render workspace does not fit") : rc;
        if (spp_pass > 1u) spp_pass = (spp_pass + 1u) / 2u;
        else slots -= 1u;
    }
    ctx->error.clear();
    int rc = PT_OK;
    char* base = (char*)ctx->work;
    float4* acc = (float4*)(base + o_acc);
    PassState st{};
    st.pray[0] = (float4*)(base + o_ray0); st.pray[1] = (float4*)(base + o_ray1);
    st.pmeta[0] = (uint4*)(base + o_meta0); st.pmeta[1] = (uint4*)(base + o_meta1);
    st.hits = (uint4*)(base + o_hits); st.tail = (float4*)(base + o_tail);
    st.queue[0] = (uint32_t*)(base + o_q0); st.queue[1] = (uint32_t*)(base + o_q1); st.miss_queue = (uint32_t*)(base + o_qm);
    st.counts = (uint32_t*)(base + o_cnt); st.verts = (float4*)(base + o_verts);
    st.npix = npix;
    st.cap = spp_pass * npix;
    st.N = slots * st.cap;

    hipStream_t s = ctx->stream;
    PTC_HIP(ctx, hipEventRecord(ctx->ev[2], s));
    PTC_HIP(ctx, hipMemsetAsync(acc, 0, (size_t)npix * sizeof(float4), s));
    PTC_HIP(ctx, hipMemsetAsync(st.counts, 0, 256, s));
    const uint32_t pix_blocks = (npix + PTC_BLOCK - 1) / PTC_BLOCK;
    // per-kernel device time (pt_stats): events from the context's pool bracket every launch group; nothing waits on them
    // until the render is over
    enum { T_TRACE = 0, T_SHADE, T_SKY, T_ACCUMULATE, T_GENERATE, T_KINDS };
    struct Interval { uint32_t kind; size_t begin, end; };
    std::vector<Interval> timed;
    size_t n_ev = 0, n_trace_launches = 0;
    int mark_rc = PT_OK;
    auto mark = [&]() -> size_t {                         // records the next pooled event on the stream, returns its index
        const int r = ptc_timed_event(ctx, n_ev);
        if (r != PT_OK && mark_rc == PT_OK) mark_rc = r;
        return n_ev++;
    };

    // The path loop runs AHEAD of the host: iteration i's kernels read the live count from the device,
    // the count is copied back asynchronously and only consulted LOOKAHEAD iterations later - to shrink
    // the grids, to fold a pass whose paths have all ended, to admit the next pass once the paths in flight
    // have thinned out, and to stop (at most LOOKAHEAD surplus iterations are enqueued at the end of the
    // render; with a zero count their kernels return at once).
    constexpr uint32_t LOOKAHEAD = 2, RING = 8;
    struct Pass { uint32_t slot, spp, paths, born; };     // born: the first iteration that traces its paths
    Pass fifo[PTC_MAX_SLOTS];                             // passes in flight, oldest first
    uint32_t n_fifo = 0, next_slot = 0;
    uint32_t generated = 0;                               // samples handed to k_generate so far
    uint32_t enq = 0, harvested = 0;                      // iterations enqueued / counts read back
    uint32_t watch_since = 0;                             // iterations >= this count the survivors of fifo[0]'s slot
    uint32_t live_ub = 0;                                 // upper bound of the live count of the next iteration to enqueue
    const uint32_t admit_below = (uint32_t)std::min<uint64_t>(st.cap, (uint64_t)((double)st.cap * ctx->tune_admit_frac));
    auto admit = [&]() {                                  // the next pass joins the queue of iteration `enq`
        const uint32_t this_spp = std::min(spp_pass, spp_count - generated);
        const uint32_t P = this_spp * npix;
        st.sample_base[next_slot] = spp_begin + generated;
        st.group[next_slot] = 1u;
        for (uint32_t g = ctx->tune_sample_group; g > 1u; g >>= 1) if (this_spp % g == 0u) { st.group[next_slot] = g; break; }
        const size_t g0 = mark();
        hipLaunchKernelGGL(k_generate, dim3((P + PTC_BLOCK - 1) / PTC_BLOCK), dim3(PTC_BLOCK), 0, s, st, *cam, seed, next_slot, P, enq % 3u, enq & 1u);
        hipLaunchKernelGGL(k_admit, dim3(1), dim3(64), 0, s, st.counts, enq % 3u, P, n_fifo == 0u ? 1u : 0u);
        timed.push_back({T_GENERATE, g0, mark()});
        if (n_fifo == 0u) watch_since = enq;
        fifo[n_fifo++] = Pass{next_slot, this_spp, P, enq};
        next_slot = (next_slot + 1u) % slots;
        generated += this_spp;
        live_ub += P;
        ctx->stats.paths += P;
        ctx->stats.segments += P;
    };
    auto fold_front = [&](bool& sky_done) {               // every path of the oldest pass has ended
        size_t m0 = mark();
        if (!sky_done) {
            const uint32_t sky_blocks = std::min<uint32_t>((st.N + PTC_BLOCK - 1) / PTC_BLOCK, (uint32_t)ctx->n_cus * 16u);
            hipLaunchKernelGGL(k_sky, dim3(sky_blocks), dim3(PTC_BLOCK), 0, s, scene->view, scene->d_sky, st, normalize_sky);
            sky_done = true;
            const size_t m1 = mark();
            timed.push_back({T_SKY, m0, m1});
            m0 = m1;
        }
        hipLaunchKernelGGL(k_accumulate, dim3((npix + PTC_BLOCK / st.group[fifo[0].slot] - 1) / (PTC_BLOCK / st.group[fifo[0].slot])), dim3(PTC_BLOCK), 0, s, st, fifo[0].slot, fifo[0].spp, acc);
        timed.push_back({T_ACCUMULATE, m0, mark()});
        for (uint32_t i = 1; i < n_fifo; ++i) fifo[i - 1] = fifo[i];
        n_fifo -= 1u;
        watch_since = enq;                                // k_shade launches from now on watch the new front
    };
    if (spp_count != 0u) admit();
    while (n_fifo != 0u) {
        if (enq - harvested <= LOOKAHEAD) {
            const size_t t0 = mark();
            rc = ptc_launch_trace(ctx, scene, (const pt_ray*)st.pray[enq & 1u], nullptr, st.queue[enq & 1u], &st.counts[2u * (enq % 3u)],
                                  live_ub, (pt_hit*)st.hits);
            if (rc != PT_OK) return rc;
            const size_t t1 = mark();
            timed.push_back({T_TRACE, t0, t1});
            n_trace_launches += 1;
            const uint32_t shade_blocks = std::max(1u, std::min<uint32_t>((live_ub + PTC_SHADE_SPAN - 1) / PTC_SHADE_SPAN, (uint32_t)ctx->n_cus * ctx->tune_shade_blocks_per_cu));
            hipLaunchKernelGGL(k_shade, dim3(shade_blocks), dim3(PTC_BLOCK), 0, s, scene->view,
                               st, enq & 1u, enq % 3u, seed, fifo[0].slot);
            timed.push_back({T_SHADE, t1, mark()});
            if (mark_rc != PT_OK) return mark_rc;
            PTC_HIP(ctx, hipMemcpyAsync(&ctx->h_counters[2u * (enq % RING)], &st.counts[2u * ((enq + 1u) % 3u)], 2 * sizeof(uint32_t),
                                        hipMemcpyDeviceToHost, s));
            PTC_HIP(ctx, hipEventRecord(ctx->count_ev[enq % RING], s));
            enq += 1;
            continue;
        }
        // read back the oldest outstanding pair (blocks only when the host is LOOKAHEAD ahead): what iteration h + 1 finds
        // in its queue, before any pass admitted at h + 1 or later
        const uint32_t h = harvested;
        PTC_HIP(ctx, hipEventSynchronize(ctx->count_ev[h % RING]));
        const uint32_t live = ctx->h_counters[2u * (h % RING)], live_watched = ctx->h_counters[2u * (h % RING) + 1u];
        harvested += 1;
        ctx->stats.segments += live;
        live_ub = live;
        for (uint32_t i = 0; i < n_fifo; ++i) if (fifo[i].born > h) live_ub += fifo[i].paths;      // admitted since
        bool sky_done = false;
        while (n_fifo != 0u && fifo[0].born <= h && (live == 0u || (h >= watch_since && live_watched == 0u))) fold_front(sky_done);
        // a path takes part in at most PTC_MAX_ITERS iterations (k_shade counts them in its state word): a pass that outlives
        // that has lost its state to something else writing the workspace - stop instead of enqueueing for ever
        if (n_fifo != 0u && h > fifo[0].born + PTC_MAX_ITERS + 8u)
            return ptc_fail(ctx, PT_ERR_HIP, "pt_render_dev: paths outlived the iteration limit (workspace corrupted?)");
        if (generated < spp_count && n_fifo < slots && live_ub <= admit_below) admit();
    }
    const float inv_total = (mode == PT_RENDER_FINAL) ? 1.0f / (float)spp_total : 0.0f;
    hipLaunchKernelGGL(k_output, dim3(pix_blocks), dim3(PTC_BLOCK), 0, s, acc, (float4*)d_xyzw, npix, mode, inv_total,
                       (float)spp_count);
    PTC_HIP(ctx, hipGetLastError());
    PTC_HIP(ctx, hipEventRecord(ctx->ev[3], s));
    PTC_HIP(ctx, hipEventSynchronize(ctx->ev[3]));
    float total = 0.0f;
    PTC_HIP(ctx, hipEventElapsedTime(&total, ctx->ev[2], ctx->ev[3]));
    if (mark_rc != PT_OK) return mark_rc;
    double by_kind[T_KINDS] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (const Interval& iv : timed) {
        float ms = 0.0f;
        PTC_HIP(ctx, hipEventElapsedTime(&ms, ctx->trace_ev[iv.begin], ctx->trace_ev[iv.end]));
        by_kind[iv.kind] += ms;
    }
    ctx->stats.total_ms = total;
    ctx->stats.trace_ms = by_kind[T_TRACE];
    ctx->stats.shade_ms = by_kind[T_SHADE];
    ctx->stats.sky_ms = by_kind[T_SKY];
    ctx->stats.accumulate_ms = by_kind[T_ACCUMULATE];
    ctx->stats.generate_ms = by_kind[T_GENERATE];
    ctx->stats.trace_rays = ctx->stats.segments;
    ctx->stats.trace_launches = (uint32_t)n_trace_launches;
    return PT_OK;
}

static int pt_render_body(pt_ctx* ctx, const pt_scene* scene, const pt_camera* cam, uint32_t spp_begin,
                         uint32_t spp_count, uint32_t spp_total, uint64_t seed, uint32_t mode, float* xyzw) {
    if (!ctx) return PT_ERR_INVALID;
    if (!scene || !cam || !xyzw) return ptc_fail(ctx, PT_ERR_INVALID, "pt_render: NULL argument");
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)cam->width * cam->height * 4 * sizeof(float);
    if (bytes == 0) return ptc_fail(ctx, PT_ERR_INVALID, "pt_render: empty image");
    int rc = ptc_reserve(ctx, &ctx->io, &ctx->io_bytes, bytes, "image staging");
    if (rc != PT_OK) return rc;
    rc = pt_render_dev(ctx, scene, cam, spp_begin, spp_count, spp_total, seed, mode, (float*)ctx->io);
    if (rc != PT_OK) return rc;
    PTC_HIP(ctx, hipMemcpy(xyzw, ctx->io, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_render(pt_ctx* ctx, const pt_scene* scene, const pt_camera* cam, uint32_t spp_begin,
                         uint32_t spp_count, uint32_t spp_total, uint64_t seed, uint32_t mode, float* xyzw) {
    return ptc_guard(ctx, "pt_render", [&] { return pt_render_body(ctx, scene, cam, spp_begin, spp_count, spp_total, seed, mode, xyzw); });
}
