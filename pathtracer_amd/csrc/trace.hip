// trace.hip — closest-hit BVH traversal + Havel–Herout ray/triangle test for gfx950.
//
// Replaces, on the device:
//   Ray::new / Ray::intersects_aabb        bvh.rs:380-410
//   BVHNode::traverse_recursive            bvh.rs:250-338   (BVH::traverse bvh.rs:358-369)
//   Triangle::intersect                    triangle.rs:82-115
//
// One ray per lane, wave64.  The traversal stack is a per-lane SHORT stack in LDS laid out
// [slot][lane] (8-byte entries, so a wave's push/pop is one conflict-free ds_write_b64 /
// ds_read_b64); it is a ring: when a lane's ring is full the oldest entry spills to a per-lane
// column in HBM ([depth][thread], coalesced) and comes back when the ring runs dry.  Binary nodes are
// fetched as 4 x 16-byte loads, wide nodes as 7, triangles as 3 x 16-byte loads per lane.
//
// Three flavours (pt_set_traversal):
//  WIDE   (default) the FAST rule over two levels of the same tree at a time: the 128-byte records of
//         device_types.h hold the four grandchildren's boxes (bit for bit the binary nodes' boxes); all four
//         are tested, sorted by entry distance (ties in binary order), the nearest is entered, the others are
//         pushed farthest first.  Half as many dependent fetches per ray as FAST.
//  FAST   near-first DFS with ONE global best distance: a subtree is entered iff best > its entry
//         distance, a leaf hit replaces the best iff t <= best (so among equal t the later
//         visited wins, as bvh.rs:299-303).  Visits a subset of what the reference visits and
//         returns the same minimum-t hit.  The comparison with a box is strict only against a hit
//         found in the near sibling's subtree (ptc_worth_entering below), which keeps equal-t ties and
//         hits a rounding error apart with the recursion: FAST agreed with it on every ray of the
//         randomised sweep, WIDE on all but 7 ray tests of 750 M.
//  EXACT  replays the recursion frame by frame: the far child of a node is entered iff the NEAR
//         SUBTREE's own result is farther than the far box (bvh.rs:285), no global pruning,
//         result = first_distance < second_distance ? first : second (bvh.rs:299-303).  Frames
//         live on the same LDS ring: an "A" frame {far ref, far tmin} waits for the near result,
//         a "B" frame {first id | tag, first distance} waits for the far result.
// FAST, WIDE and EXACT run as persistent kernels that hand the next ray to a lane as soon as it is done
// (RayTickets); PTC_TRACE_VARIANT=0 selects the older one-batch-per-wave kernels of FAST and EXACT.
//
// All float arithmetic is written in the reference's operation order and the file is compiled
// with -ffp-contract=off: no FMA, IEEE division, minNum/maxNum for f32::min/max.

#include <hip/hip_runtime.h>

#include "device_types.h"
#include "../../include/pt_math.h"

#define PTC_BLOCK 256
#ifndef PTC_SHARDS
#define PTC_SHARDS 8u             /* ticket words / contiguous ray shards per launch (power of two not required) */
#endif
#define PTC_FIRST 64u              /* rays of a wave's static first block (one per lane) */

namespace {

struct DRay { float ox, oy, oz, dx, dy, dz, ix, iy, iz; };

__device__ __forceinline__ DRay make_ray(float4 o, float4 d) {
    DRay r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.ix = 1.0f / d.x; r.iy = 1.0f / d.y; r.iz = 1.0f / d.z;      // bvh.rs:384
    return r;
}

// Ray::intersects_aabb (bvh.rs:390-410); returns hit flag, entry distance in tmin
__device__ __forceinline__ bool slab(const DRay& r, float bx0, float by0, float bz0, float bx1,
                                     float by1, float bz1, float& tmin_out) {
    float t1x = (bx0 - r.ox) * r.ix, t2x = (bx1 - r.ox) * r.ix;
    float t1y = (by0 - r.oy) * r.iy, t2y = (by1 - r.oy) * r.iy;
    float t1z = (bz0 - r.oz) * r.iz, t2z = (bz1 - r.oz) * r.iz;
    float tmin = pt_fmin(t1x, t2x);
    float tmax = pt_fmax(t1x, t2x);
    tmin = pt_fmax(tmin, pt_fmin(t1y, t2y));
    tmax = pt_fmin(tmax, pt_fmax(t1y, t2y));
    tmin = pt_fmax(tmin, pt_fmin(t1z, t2z));
    tmax = pt_fmin(tmax, pt_fmax(t1z, t2z));
    tmin = pt_fmax(tmin, 0.0f);
    tmin_out = tmin;
    return tmax >= tmin;
}

// The same test split for the 4-box records: slab_axis2 does one axis of two boxes (packed f32
// subtract and multiply: one rounding per half, as in `slab`), slab_close the min/max chain of one box.
typedef float ptc_f2 __attribute__((ext_vector_type(2)));
struct PairF { ptc_f2 t1, t2; };
__device__ __forceinline__ PairF slab_axis2(float lo_a, float lo_b, float hi_a, float hi_b, float o, float inv) {
    const ptc_f2 lo = {lo_a, lo_b}, hi = {hi_a, hi_b}, oo = {o, o}, ii = {inv, inv};
    PairF p;
    p.t1 = (lo - oo) * ii;
    p.t2 = (hi - oo) * ii;
    return p;
}
__device__ __forceinline__ bool slab_close(float t1x, float t2x, float t1y, float t2y, float t1z, float t2z,
                                           float& tmin_out) {
    float tmin = pt_fmin(t1x, t2x);
    float tmax = pt_fmax(t1x, t2x);
    tmin = pt_fmax(tmin, pt_fmin(t1y, t2y));
    tmax = pt_fmin(tmax, pt_fmax(t1y, t2y));
    tmin = pt_fmax(tmin, pt_fmin(t1z, t2z));
    tmax = pt_fmin(tmax, pt_fmax(t1z, t2z));
    tmin = pt_fmax(tmin, 0.0f);
    tmin_out = tmin;
    return tmax >= tmin;
}

// Triangle::intersect with max_distance = INFINITY (bvh.rs:332), branch-free until the verdict
__device__ __forceinline__ bool tri_test(const DRay& r, float4 t0, float4 t1, float4 t2,
                                         float& dist, float& uo, float& vo) {
    float det = (r.dx * t0.x + r.dy * t0.y) + r.dz * t0.z;                   // triangle.rs:84
    float t = t0.w - ((r.ox * t0.x + r.oy * t0.y) + r.oz * t0.z);            // :89
    float m = pt_inf() * det - t;                                            // :91
    float px = r.ox * det + r.dx * t;                                        // :95
    float py = r.oy * det + r.dy * t;
    float pz = r.oz * det + r.dz * t;
    float u = ((px * t1.x + py * t1.y) + pz * t1.z) + det * t1.w;            // :96
    float v = ((px * t2.x + py * t2.y) + pz * t2.z) + det * t2.w;            // :101
    float du = det - u;
    float duv = du - v;                                                      // (det - u) - v, :102
    uint32_t bad = (pt_f2u(m) ^ pt_f2u(t)) | (pt_f2u(du) ^ pt_f2u(u)) | (pt_f2u(duv) ^ pt_f2u(v));
    float inv = 1.0f / det;                                                  // :106
    float d = t * inv;                                                       // :108
    bool ok = (det != 0.0f) && ((bad >> 31) == 0u);
    // relative_eq!(distance, 0.0): equal, or |distance| <= f32::EPSILON      :110
    ok = ok && !(d == 0.0f || pt_fabs(d) <= 1.1920928955078125e-7f);
    dist = d;
    uo = u * inv;
    vo = v * inv;
    return ok;
}

// ---- per-lane ring stack: LDS window over an HBM column -----------------------------------------
// `ring` is the block's __shared__ array, slot-major ([slot][thread]); the column is
// [depth][thread] in HBM.  LDS and HBM accesses are kept in separate, differently-flavoured
// branches on purpose: if both sides were plain loads the compiler would select between the two
// ADDRESSES and emit one flat_load, which is far slower than ds_read_b64.
__device__ __forceinline__ uint32_t ptc_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

template <int S>
struct LaneStack {
    uint2* spill;        // &column[0][gtid]; depth stride = spill_stride
    uint32_t spill_stride;
    uint32_t idx;        // element index of the next free ring slot: slot * PTC_BLOCK + threadIdx.x
    uint32_t cnt;        // entries in LDS (the newest `cnt` of the stack)
    uint32_t base;       // entries below them, in HBM

    __device__ __forceinline__ void reset() { idx = threadIdx.x; cnt = 0u; base = 0u; }
    __device__ __forceinline__ bool empty() const { return (cnt | base) == 0u; }
    __device__ __forceinline__ void push(uint2* __restrict__ ring, uint32_t a, uint32_t b) {
        if (__builtin_expect(cnt == (uint32_t)S, 0)) {      // ring full: the next free slot holds the oldest entry
            const uint2 old = ring[idx];
            __builtin_nontemporal_store(old.x, &spill[(size_t)base * spill_stride].x);
            __builtin_nontemporal_store(old.y, &spill[(size_t)base * spill_stride].y);
            base += 1u;
            cnt -= 1u;
        }
        ring[idx] = make_uint2(a, b);
        idx += PTC_BLOCK;
        idx = ptc_umin(idx, idx - (uint32_t)(S * PTC_BLOCK));      // wraps: idx - S*BLOCK underflows unless idx >= S*BLOCK
        cnt += 1u;
    }
    __device__ __forceinline__ uint2 pop(const uint2* __restrict__ ring) {     // caller checks !empty()
        // the LDS read is unconditional (its slot always exists) so that it stays a ds_read_b64;
        // the rare HBM refill overrides it
        const uint32_t prev = ptc_umin(idx - PTC_BLOCK, idx + (uint32_t)((S - 1) * PTC_BLOCK));
        uint2 e = ring[prev];
        if (__builtin_expect(cnt == 0u, 0)) {
            base -= 1u;
            // volatile: keeps this a separate global load (otherwise LLVM selects between the LDS
            // and HBM ADDRESSES and emits one slow flat_load for every pop)
            const volatile uint32_t* col = reinterpret_cast<const volatile uint32_t*>(&spill[(size_t)base * spill_stride]);
            e.x = col[0];
            e.y = col[1];
        } else {
            cnt -= 1u;
            idx = prev;
        }
        return e;
    }
};

__device__ __forceinline__ void load_node(const float4* nodes, uint32_t idx, float4& q0, float4& q1,
                                          float4& q2, float4& q3) {
    const float4* p = nodes + (size_t)idx * 4u;
    q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
}

// ---- dynamic ray reservation of the persistent kernels (wave-uniform state) ---------------------
// The batch is cut into PTC_SHARDS contiguous shards, each with its own ticket word; a wave works on the shard
// of its block label (blockIdx % PTC_SHARDS: blocks with the same label tend to share an XCD, hence an L2) and
// moves on to the next shard when its own runs dry.  A wave's FIRST block is static (its rank among the waves
// of its label): otherwise every wave of the grid hits a ticket word at launch, and one word takes only ~90
// returning atomics per microsecond.
struct RayTickets {
    uint32_t blk_next, blk_end;      // unassigned part of the reserved block
    uint32_t shard, shards_done, shard_len, waves_per_label;
    bool exhausted;                  // every shard's counter ran past its end
    bool first_block;

    __device__ __forceinline__ void init(uint32_t n) {
        blk_next = blk_end = 0u;
        exhausted = false;
        first_block = true;
        shard = blockIdx.x % PTC_SHARDS;
        shards_done = 0u;
        // shard length: a multiple of the static block, so that static and dynamic reservations tile a shard exactly
        shard_len = ((n + PTC_SHARDS * PTC_FIRST - 1u) / (PTC_SHARDS * PTC_FIRST)) * PTC_FIRST;
        waves_per_label = (gridDim.x / PTC_SHARDS) * (PTC_BLOCK / 64u);            // grid is a multiple of PTC_SHARDS
    }
    // Hands the next rays to the wave's idle lanes (ballot `idle`, `n_idle` of them): true and the batch position
    // `k` for a lane that got one.  Reserves further blocks of `ticket_size` rays as needed.
    __device__ __forceinline__ bool take(unsigned long long idle, uint32_t n_idle, uint32_t lane, bool lane_idle, uint32_t n,
                                         uint32_t* __restrict__ ticket, uint32_t ticket_size, uint32_t& k) {
        while (blk_next == blk_end && !exhausted) {
            uint32_t off = 0u, size = ticket_size;
            if (first_block) {
                off = ((blockIdx.x / PTC_SHARDS) * (PTC_BLOCK / 64u) + (threadIdx.x >> 6)) * PTC_FIRST;
                size = PTC_FIRST;
                first_block = false;
            } else {
                if (lane == 0u) off = atomicAdd(ticket + shard * 64u, ticket_size);
                off = __builtin_amdgcn_readfirstlane(off) + waves_per_label * PTC_FIRST;
            }
            const uint32_t shard_begin = shard * shard_len;
            if (off >= shard_len || shard_begin + off >= n) {
                shards_done += 1u;
                if (shards_done == PTC_SHARDS) { exhausted = true; blk_next = blk_end = 0u; }
                else shard = (shard + 1u == PTC_SHARDS) ? 0u : shard + 1u;
            } else {
                const uint32_t shard_end = (n - shard_begin < shard_len) ? n : shard_begin + shard_len;
                blk_next = shard_begin + off;
                blk_end = (shard_end - blk_next < size) ? shard_end : blk_next + size;
            }
        }
        const uint32_t avail = blk_end - blk_next;
        const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        k = blk_next + rank;
        blk_next += (n_idle < avail) ? n_idle : avail;
        return lane_idle && rank < avail;
    }
};

#define PTC_IDLE 0xFFFFFFFFu   /* "needs a pop": leaf bit set, never a valid triangle index */

// FAST / WIDE pruning rule.  The recursion never compares a box with a hit from ANOTHER subtree: it enters every box
// the ray meets, and skips the far child only if the near child's own result is not farther than the far box
// (`first.distance > far_min`, strict, bvh.rs:285).  These kernels prune against the best hit of the whole ray, which
// in exact arithmetic loses nothing; to stay with the recursion where distances are EQUAL (coplanar triangles, flat
// boxes entered exactly at the triangle's t) the comparison is strict only when the best hit was found after the
// entry was pushed - i.e. inside the near sibling's subtree, the one case in which the recursion compares at all - and
// non-strict otherwise.  The parity of the number of accepted hits travels with every stack entry for that, in bit 30
// of the reference (free outside EXACT); `tagbit` is that parity, kept in place.  Distances are non-negative, so
// their bit patterns order like the floats and "strict" is "one less": enter iff bits(best) - 1 >= bits(entry).
// Against a hit from another subtree FAST's comparison also gives the box 2 ulp: a box's entry distance,
// (min - o) * inv, and the distance of the triangle inside it, (d - o.n) / det, are rounded independently, and the
// entry of a flat box can lie an ulp or two above the t of its own triangle.
// The slack is FAST's alone: on WIDE it cost 4.5 % of the integrator's traversal rate (measured), and WIDE's remaining
// differences from the recursion are ties of another kind.
template <bool WIDE>
__device__ __forceinline__ bool ptc_may_hold_a_hit(float entry, float best_t) {     // nothing found since: no sibling to compare
    if constexpr (WIDE) return best_t >= entry;
    else return (int32_t)__float_as_uint(entry) <= (int32_t)(__float_as_uint(best_t) + 2u);
}
template <bool WIDE>
__device__ __forceinline__ bool ptc_worth_entering(uint2 e, float best_t, uint32_t tagbit) {
    if constexpr (WIDE) {
        const uint32_t changed = ((e.x ^ tagbit) >> 30) & 1u;
        return (int32_t)(__float_as_uint(best_t) - changed) >= (int32_t)e.y;
    } else {
        const bool changed = ((e.x ^ tagbit) & PTC_FRAME_BIT) != 0u;
        const uint32_t limit = __float_as_uint(best_t) + (changed ? 0xFFFFFFFFu : 2u);
        return (int32_t)limit >= (int32_t)e.y;
    }
}

// FAST.  Every loop iteration runs three predicated phases, so a lane can visit a node, test the
// leaf it descends into and pop its next subtree in ONE trip; the wave's trip count is then about
// the largest per-lane node count instead of the sum of node, leaf and pop steps.
template <int S, bool QUEUED>
__global__ __launch_bounds__(PTC_BLOCK) void k_trace_fast(DevSceneView sc, const float4* __restrict__ rays,
                                                         const uint32_t* __restrict__ start,
                                                         const uint32_t* __restrict__ queue,
                                                         const uint32_t* __restrict__ count_ptr, uint32_t n,
                                                         uint4* __restrict__ out, uint2* __restrict__ spill,
                                                         uint32_t total_threads) {
    __shared__ uint2 ring[S * PTC_BLOCK];
    const uint32_t gtid = blockIdx.x * PTC_BLOCK + threadIdx.x;
    if (count_ptr) n = *count_ptr;
    LaneStack<S> st;
    st.spill = spill + gtid;
    st.spill_stride = total_threads;

    for (uint32_t k = gtid; k < n; k += total_threads) {
        const uint32_t i = QUEUED ? queue[k] : k;
        const float4 ro = rays[2 * (size_t)i];
        const DRay r = make_ray(ro, rays[2 * (size_t)i + 1]);
        // QUEUED (the integrator's form): the start triangle rides in the ray's pad0
        const uint32_t skip = QUEUED ? __float_as_uint(ro.w) : (start ? start[i] : PT_NO_TRIANGLE);
        float best_t = pt_inf(), best_u = 0.0f, best_v = 0.0f;
        uint32_t best_id = PT_NO_TRIANGLE, tagbit = 0u;
        st.reset();
        uint32_t cur = sc.root_ref;
        bool alive = sc.n_tris != 0u;
        while (alive) {
            // ---- phase 1: internal node (bvh.rs:259-325) ----
            if (!(cur & PTC_LEAF_BIT)) {
                float4 q0, q1, q2, q3;
                load_node(sc.nodes, cur, q0, q1, q2, q3);
                float lt, rt;
                bool lh = slab(r, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, lt);    // bvh.rs:260
                bool rh = slab(r, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, rt);    // bvh.rs:261
                lh = lh && ptc_may_hold_a_hit<false>(lt, best_t);
                rh = rh && ptc_may_hold_a_hit<false>(rt, best_t);
                const uint32_t lref = __float_as_uint(q0.w), rref = __float_as_uint(q1.w);
                const bool left_first = lt < rt;                              // bvh.rs:266
                if (lh && rh) st.push(ring, (left_first ? rref : lref) | tagbit, __float_as_uint(left_first ? rt : lt));
                cur = (lh && (left_first || !rh)) ? lref : (rh ? rref : PTC_IDLE);
            }
            // ---- phase 2: leaf (bvh.rs:327-336) ----
            if ((cur & PTC_LEAF_BIT) && cur != PTC_IDLE) {
                const uint32_t tri = cur & PTC_INDEX_MASK;
                if (tri != skip) {                                            // bvh.rs:329-330
                    const float4* tp = sc.tris + (size_t)tri * 3u;
                    float d, u, v;
                    if (tri_test(r, tp[0], tp[1], tp[2], d, u, v) && d <= best_t) {
                        best_t = d; best_u = u; best_v = v; best_id = tri;
                        tagbit ^= PTC_FRAME_BIT;
                    }
                }
                cur = PTC_IDLE;
            }
            // ---- phase 3: pop until something is still worth entering ----
            while (cur == PTC_IDLE) {
                if (st.empty()) { alive = false; break; }
                const uint2 e = st.pop(ring);
                if (ptc_worth_entering<false>(e, best_t, tagbit)) cur = e.x & ~PTC_FRAME_BIT;
            }
        }
        out[i] = make_uint4(best_id, __float_as_uint(best_id == PT_NO_TRIANGLE ? 0.0f : best_t),
                            __float_as_uint(best_u), __float_as_uint(best_v));
    }
}

// FAST, persistent form.  Traversal lengths have a heavy tail (soup-1M primaries: mean 51 node
// visits per ray, but the slowest of 64 neighbouring rays needs 141), so a wave that waits for its
// slowest lane keeps only ~1/3 of its lanes busy.  Here a lane that finishes its ray takes the next
// one at once: each wave reserves blocks of `ticket_size` (64) consecutive rays from a global ticket counter (one
// atomic per block, fetched by lane 0 and broadcast) and hands them to idle lanes by ballot +
// popcount rank, whenever at least `refill` lanes are idle.  Lanes only idle during the final drain.
//
// WIDE (PT_TRAVERSE_WIDE): the same loop over the two-level records of device_types.h.  A trip tests
// four boxes (the binary node's grandchildren, same arithmetic per box), sorts the ones worth entering
// by entry distance, descends into the nearest and pushes the others farthest first.  A ray needs
// about half as many dependent node fetches, which is what bounds the binary kernel.
template <int S, bool QUEUED, bool WIDE>
__global__ __launch_bounds__(PTC_BLOCK) void k_trace_fast_persistent(
    DevSceneView sc, const float4* __restrict__ rays, const uint32_t* __restrict__ start,
    const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t n,
    uint4* __restrict__ out, uint2* __restrict__ spill, uint32_t total_threads, uint32_t* __restrict__ ticket,
    uint32_t* __restrict__ ticket_next, uint32_t refill, uint32_t ticket_size) {
    __shared__ uint2 ring[S * PTC_BLOCK];
    const uint32_t gtid = blockIdx.x * PTC_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (count_ptr) n = *count_ptr;
    // two sets of ticket words alternate between launches; this launch clears the set the NEXT launch uses
    // (the previous launch, its last user, has completed: launches are stream-ordered)
    if (gtid < PTC_SHARDS) ticket_next[gtid * 64u] = 0u;
    LaneStack<S> st;
    st.spill = spill + gtid;
    st.spill_stride = total_threads;
    st.reset();
    if (sc.n_tris == 0u) {          // empty BVH: every ray misses (bvh.rs:364-368)
        for (uint32_t k = gtid; k < n; k += total_threads) out[QUEUED ? queue[k] : k] = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u);
        return;
    }

    RayTickets tk;
    tk.init(n);
    bool has_ray = false;
    DRay r = {};
    uint32_t ray_i = 0u, skip = PT_NO_TRIANGLE, cur = PTC_IDLE, best_id = PT_NO_TRIANGLE, tagbit = 0u;
    float best_t = pt_inf(), best_u = 0.0f, best_v = 0.0f;

    for (;;) {
        // ---- refill ----
        const unsigned long long idle = __ballot(!has_ray);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (!tk.exhausted && (n_idle >= refill || n_idle == 64u)) {
            uint32_t k;
            if (tk.take(idle, n_idle, lane, !has_ray, n, ticket, ticket_size, k)) {
                ray_i = QUEUED ? queue[k] : k;
                const float4 ro = rays[2 * (size_t)ray_i];
                r = make_ray(ro, rays[2 * (size_t)ray_i + 1]);
                // QUEUED (the integrator's form): the start triangle rides in the ray's pad0
                skip = QUEUED ? __float_as_uint(ro.w) : (start ? start[ray_i] : PT_NO_TRIANGLE);
                best_t = pt_inf(); best_u = 0.0f; best_v = 0.0f; best_id = PT_NO_TRIANGLE;
                tagbit = 0u;
                st.reset();
                cur = sc.root_ref;
                has_ray = true;
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (tk.exhausted) break;
            continue;                      // block ran dry mid-refill: reserve another one
        }
        if (has_ray) {
            // ---- phase 1: internal node (bvh.rs:259-325) ----
            if (!(cur & PTC_LEAF_BIT)) {
                if constexpr (WIDE) {
                    const float4* p = sc.wide + (size_t)cur * 8u;
                    const float4 lox = p[0], loy = p[1], loz = p[2], hix = p[3], hiy = p[4], hiz = p[5], rw = p[6];
                    float k0, k1, k2, k3;
                    uint32_t r0 = __float_as_uint(rw.x), r1 = __float_as_uint(rw.y), r2 = __float_as_uint(rw.z), r3 = __float_as_uint(rw.w);
                    // the 24 subtractions and 24 products of the four slab tests, two boxes per packed
                    // instruction (v_pk_add_f32 / v_pk_mul_f32 round each half like the scalar forms)
                    const PairF ax = slab_axis2(lox.x, lox.y, hix.x, hix.y, r.ox, r.ix), bx = slab_axis2(lox.z, lox.w, hix.z, hix.w, r.ox, r.ix);
                    const PairF ay = slab_axis2(loy.x, loy.y, hiy.x, hiy.y, r.oy, r.iy), by = slab_axis2(loy.z, loy.w, hiy.z, hiy.w, r.oy, r.iy);
                    const PairF az = slab_axis2(loz.x, loz.y, hiz.x, hiz.y, r.oz, r.iz), bz = slab_axis2(loz.z, loz.w, hiz.z, hiz.w, r.oz, r.iz);
                    const bool h0 = slab_close(ax.t1.x, ax.t2.x, ay.t1.x, ay.t2.x, az.t1.x, az.t2.x, k0) && ptc_may_hold_a_hit<true>(k0, best_t);
                    const bool h1 = slab_close(ax.t1.y, ax.t2.y, ay.t1.y, ay.t2.y, az.t1.y, az.t2.y, k1) && ptc_may_hold_a_hit<true>(k1, best_t);
                    const bool h2 = slab_close(bx.t1.x, bx.t2.x, by.t1.x, by.t2.x, bz.t1.x, bz.t2.x, k2) && ptc_may_hold_a_hit<true>(k2, best_t);
                    const bool h3 = slab_close(bx.t1.y, bx.t2.y, by.t1.y, by.t2.y, bz.t1.y, bz.t2.y, k3) && ptc_may_hold_a_hit<true>(k3, best_t);
                    k0 = h0 ? k0 : pt_inf(); k1 = h1 ? k1 : pt_inf(); k2 = h2 ? k2 : pt_inf(); k3 = h3 ? k3 : pt_inf();
                    // Ascending entry distance by odd-even transposition: six exchanges of NEIGHBOURS with a strict
                    // compare, hence stable.  The record holds the children in REVERSE binary order (RR, RL, LR, LL), so
                    // boxes with equal entry distance are entered right before left, as bvh.rs:266-270 does at every
                    // binary node (`l_min < r_min` else right first): coincident geometry resolves as in the recursion.
#define PTC_CSWAP(ka, kb, ra, rb) { const bool sw = ka > kb; const float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; \
                                    const uint32_t tr = sw ? rb : ra; rb = sw ? ra : rb; ra = tr; }
                    PTC_CSWAP(k0, k1, r0, r1) PTC_CSWAP(k2, k3, r2, r3) PTC_CSWAP(k1, k2, r1, r2)
                    PTC_CSWAP(k0, k1, r0, r1) PTC_CSWAP(k2, k3, r2, r3) PTC_CSWAP(k1, k2, r1, r2)
#undef PTC_CSWAP
                    if (k3 < pt_inf()) st.push(ring, r3 | tagbit, __float_as_uint(k3));
                    if (k2 < pt_inf()) st.push(ring, r2 | tagbit, __float_as_uint(k2));
                    if (k1 < pt_inf()) st.push(ring, r1 | tagbit, __float_as_uint(k1));
                    cur = (k0 < pt_inf()) ? r0 : PTC_IDLE;
                } else {
                    float4 q0, q1, q2, q3;
                    load_node(sc.nodes, cur, q0, q1, q2, q3);
                    float lt, rt;
                    bool lh = slab(r, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, lt);
                    bool rh = slab(r, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, rt);
                    lh = lh && ptc_may_hold_a_hit<false>(lt, best_t);
                    rh = rh && ptc_may_hold_a_hit<false>(rt, best_t);
                    const uint32_t lref = __float_as_uint(q0.w), rref = __float_as_uint(q1.w);
                    const bool left_first = lt < rt;                              // bvh.rs:266
                    if (lh && rh) st.push(ring, (left_first ? rref : lref) | tagbit, __float_as_uint(left_first ? rt : lt));
                    cur = (lh && (left_first || !rh)) ? lref : (rh ? rref : PTC_IDLE);
                }
            }
            // ---- phase 2: leaf (bvh.rs:327-336) ----
            if ((cur & PTC_LEAF_BIT) && cur != PTC_IDLE) {
                const uint32_t tri = cur & PTC_INDEX_MASK;
                if (tri != skip) {
                    const float4* tp = sc.tris + (size_t)tri * 3u;
                    float d, u, v;
                    if (tri_test(r, tp[0], tp[1], tp[2], d, u, v) && d <= best_t) {
                        best_t = d; best_u = u; best_v = v; best_id = tri;
                        tagbit ^= PTC_FRAME_BIT;
                    }
                }
                cur = PTC_IDLE;
            }
            // ---- phase 3: pop; an empty stack ends the ray ----
            while (cur == PTC_IDLE) {
                if (st.empty()) {
                    out[ray_i] = make_uint4(best_id, __float_as_uint(best_id == PT_NO_TRIANGLE ? 0.0f : best_t),
                                            __float_as_uint(best_u), __float_as_uint(best_v));
                    has_ray = false;
                    break;
                }
                const uint2 e = st.pop(ring);
                if (ptc_worth_entering<WIDE>(e, best_t, tagbit)) cur = e.x & ~PTC_FRAME_BIT;
            }
        }
    }
}

// EXACT: frame-by-frame replay of traverse_recursive.
template <int S, bool QUEUED>
__global__ __launch_bounds__(PTC_BLOCK) void k_trace_exact(DevSceneView sc, const float4* __restrict__ rays,
                                                          const uint32_t* __restrict__ start,
                                                          const uint32_t* __restrict__ queue,
                                                          const uint32_t* __restrict__ count_ptr, uint32_t n,
                                                          uint4* __restrict__ out, uint2* __restrict__ spill,
                                                          uint32_t total_threads) {
    __shared__ uint2 ring[S * PTC_BLOCK];
    const uint32_t gtid = blockIdx.x * PTC_BLOCK + threadIdx.x;
    if (count_ptr) n = *count_ptr;
    LaneStack<S> st;
    st.spill = spill + gtid;
    st.spill_stride = total_threads;

    for (uint32_t k = gtid; k < n; k += total_threads) {
        const uint32_t i = QUEUED ? queue[k] : k;
        const float4 ro = rays[2 * (size_t)i];
        const DRay r = make_ray(ro, rays[2 * (size_t)i + 1]);
        // QUEUED (the integrator's form): the start triangle rides in the ray's pad0
        const uint32_t skip = QUEUED ? __float_as_uint(ro.w) : (start ? start[i] : PT_NO_TRIANGLE);
        // value being returned up the (virtual) call stack: None = {NONE_ID, +inf}
        uint32_t ret_id = PTC_NONE_ID;
        float ret_t = pt_inf();
        st.reset();
        uint32_t cur = sc.root_ref;
        bool descending = sc.n_tris != 0u;
        bool alive = descending;
        while (alive) {
            if (descending) {
                if (cur & PTC_LEAF_BIT) {                                     // bvh.rs:327-336
                    const uint32_t tri = cur & PTC_INDEX_MASK;
                    ret_id = PTC_NONE_ID; ret_t = pt_inf();
                    if (tri != skip) {
                        const float4* tp = sc.tris + (size_t)tri * 3u;
                        float d, u, v;
                        if (tri_test(r, tp[0], tp[1], tp[2], d, u, v)) { ret_id = tri; ret_t = d; }
                    }
                    descending = false;
                } else {
                    float4 q0, q1, q2, q3;
                    load_node(sc.nodes, cur, q0, q1, q2, q3);
                    float lt, rt;
                    const bool lh = slab(r, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, lt);
                    const bool rh = slab(r, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, rt);
                    const uint32_t lref = __float_as_uint(q0.w), rref = __float_as_uint(q1.w);
                    if (lh && rh) {                                           // bvh.rs:264-270
                        const bool left_first = lt < rt;
                        // A frame: far ref (tag bit clear) + far entry distance
                        st.push(ring, left_first ? rref : lref, __float_as_uint(left_first ? rt : lt));
                        cur = left_first ? lref : rref;
                    } else if (lh) {
                        cur = lref;                                           // bvh.rs:308-315
                    } else if (rh) {
                        cur = rref;                                           // bvh.rs:316-323
                    } else {
                        ret_id = PTC_NONE_ID; ret_t = pt_inf();               // bvh.rs:324
                        descending = false;
                    }
                }
            } else {
                if (st.empty()) { alive = false; break; }
                const uint2 e = st.pop(ring);
                const bool is_b = ((e.x & PTC_FRAME_BIT) != 0u) && ((e.x & PTC_LEAF_BIT) == 0u);
                if (!is_b) {
                    // near subtree returned `ret`; bvh.rs:280-285
                    const float first_distance = (ret_id != PTC_NONE_ID) ? ret_t : pt_inf();
                    if (first_distance > __uint_as_float(e.y)) {
                        st.push(ring, PTC_FRAME_BIT | ret_id, __float_as_uint(ret_t));   // B frame keeps `first`
                        cur = e.x;
                        descending = true;
                    }
                    // else: return first_intersection unchanged (bvh.rs:304-306)
                } else {
                    // far subtree returned `ret`; bvh.rs:294-303
                    const uint32_t f_id = e.x & PTC_INDEX_MASK;
                    const float f_t = __uint_as_float(e.y);
                    const float fd = (f_id != PTC_NONE_ID) ? f_t : pt_inf();
                    const float sd = (ret_id != PTC_NONE_ID) ? ret_t : pt_inf();
                    if (fd < sd) { ret_id = f_id; ret_t = f_t; }
                }
            }
        }
        uint4 o = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u);
        if (ret_id != PTC_NONE_ID) {
            // u, v of the winner: the same arithmetic on the same inputs gives the same bits
            const float4* tp = sc.tris + (size_t)ret_id * 3u;
            float d, u, v;
            tri_test(r, tp[0], tp[1], tp[2], d, u, v);
            o = make_uint4(ret_id, __float_as_uint(d), __float_as_uint(u), __float_as_uint(v));
        }
        out[i] = o;
    }
}

// EXACT, persistent form: the same frames, with the ray reservation of the FAST kernel (a lane whose recursion has
// returned takes the next ray at once) and the three phases per trip (node step, leaf, returns).
template <int S, bool QUEUED>
__global__ __launch_bounds__(PTC_BLOCK) void k_trace_exact_persistent(
    DevSceneView sc, const float4* __restrict__ rays, const uint32_t* __restrict__ start,
    const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t n,
    uint4* __restrict__ out, uint2* __restrict__ spill, uint32_t total_threads, uint32_t* __restrict__ ticket,
    uint32_t* __restrict__ ticket_next, uint32_t refill, uint32_t ticket_size) {
    __shared__ uint2 ring[S * PTC_BLOCK];
    const uint32_t gtid = blockIdx.x * PTC_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (count_ptr) n = *count_ptr;
    if (gtid < PTC_SHARDS) ticket_next[gtid * 64u] = 0u;      // see k_trace_fast_persistent
    LaneStack<S> st;
    st.spill = spill + gtid;
    st.spill_stride = total_threads;
    st.reset();
    if (sc.n_tris == 0u) {          // empty BVH: every ray misses (bvh.rs:364-368)
        for (uint32_t k = gtid; k < n; k += total_threads) out[QUEUED ? queue[k] : k] = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u);
        return;
    }
    RayTickets tk;
    tk.init(n);
    bool has_ray = false, descending = false;
    DRay r = {};
    uint32_t ray_i = 0u, skip = PT_NO_TRIANGLE, cur = 0u;
    uint32_t ret_id = PTC_NONE_ID;       // value being returned up the (virtual) call stack: None = {NONE_ID, +inf}
    float ret_t = pt_inf();

    for (;;) {
        const unsigned long long idle = __ballot(!has_ray);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (!tk.exhausted && (n_idle >= refill || n_idle == 64u)) {
            uint32_t k;
            if (tk.take(idle, n_idle, lane, !has_ray, n, ticket, ticket_size, k)) {
                ray_i = QUEUED ? queue[k] : k;
                const float4 ro = rays[2 * (size_t)ray_i];
                r = make_ray(ro, rays[2 * (size_t)ray_i + 1]);
                skip = QUEUED ? __float_as_uint(ro.w) : (start ? start[ray_i] : PT_NO_TRIANGLE);
                ret_id = PTC_NONE_ID; ret_t = pt_inf();
                st.reset();
                cur = sc.root_ref;
                descending = true;
                has_ray = true;
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (tk.exhausted) break;
            continue;
        }
        if (has_ray) {
            // ---- phase 1: internal node (bvh.rs:259-325) ----
            if (descending && !(cur & PTC_LEAF_BIT)) {
                float4 q0, q1, q2, q3;
                load_node(sc.nodes, cur, q0, q1, q2, q3);
                float lt, rt;
                const bool lh = slab(r, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, lt);
                const bool rh = slab(r, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, rt);
                const uint32_t lref = __float_as_uint(q0.w), rref = __float_as_uint(q1.w);
                const bool left_first = lt < rt;                              // bvh.rs:266
                // A frame: far ref (tag bit clear) + far entry distance
                if (lh && rh) st.push(ring, left_first ? rref : lref, __float_as_uint(left_first ? rt : lt));
                cur = (lh && (left_first || !rh)) ? lref : rref;             // bvh.rs:264-323
                if (!(lh || rh)) { ret_id = PTC_NONE_ID; ret_t = pt_inf(); descending = false; }   // bvh.rs:324
            }
            // ---- phase 2: leaf (bvh.rs:327-336) ----
            if (descending && (cur & PTC_LEAF_BIT)) {
                const uint32_t tri = cur & PTC_INDEX_MASK;
                ret_id = PTC_NONE_ID; ret_t = pt_inf();
                if (tri != skip) {
                    const float4* tp = sc.tris + (size_t)tri * 3u;
                    float d, u, v;
                    if (tri_test(r, tp[0], tp[1], tp[2], d, u, v)) { ret_id = tri; ret_t = d; }
                }
                descending = false;
            }
            // ---- phase 3: return through the frames until one sends the lane down again ----
            while (!descending) {
                if (st.empty()) {
                    uint4 o = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u);
                    if (ret_id != PTC_NONE_ID) {
                        // u, v of the winner: the same arithmetic on the same inputs gives the same bits
                        const float4* tp = sc.tris + (size_t)ret_id * 3u;
                        float d, u, v;
                        tri_test(r, tp[0], tp[1], tp[2], d, u, v);
                        o = make_uint4(ret_id, __float_as_uint(d), __float_as_uint(u), __float_as_uint(v));
                    }
                    out[ray_i] = o;
                    has_ray = false;
                    break;
                }
                const uint2 e = st.pop(ring);
                const bool is_b = ((e.x & PTC_FRAME_BIT) != 0u) && ((e.x & PTC_LEAF_BIT) == 0u);
                if (!is_b) {
                    // near subtree returned `ret`; bvh.rs:280-285
                    const float first_distance = (ret_id != PTC_NONE_ID) ? ret_t : pt_inf();
                    if (first_distance > __uint_as_float(e.y)) {
                        st.push(ring, PTC_FRAME_BIT | ret_id, __float_as_uint(ret_t));   // B frame keeps `first`
                        cur = e.x;
                        descending = true;
                    }
                    // else: return first_intersection unchanged (bvh.rs:304-306)
                } else {
                    // far subtree returned `ret`; bvh.rs:294-303
                    const uint32_t f_id = e.x & PTC_INDEX_MASK;
                    const float f_t = __uint_as_float(e.y);
                    const float fd = (f_id != PTC_NONE_ID) ? f_t : pt_inf();
                    const float sd = (ret_id != PTC_NONE_ID) ? ret_t : pt_inf();
                    if (fd < sd) { ret_id = f_id; ret_t = f_t; }
                }
            }
        }
    }
}

}  // namespace

// Leaf refs carry PTC_LEAF_BIT with PTC_FRAME_BIT clear (indices < 2^30), internal refs have both
// clear, so an A frame never looks like a B frame.

int ptc_launch_trace(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays, const uint32_t* d_start,
                     const uint32_t* d_queue, const uint32_t* d_count, uint32_t n, pt_hit* d_out) {
    if (n == 0) return PT_OK;
    if (n > 0xF0000000u) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "more than 0xF0000000 rays in one batch");
    const DevSceneView& v = scene->view;
    // resident grid: waves_per_simd blocks of 256 per CU at most; fewer when the batch is small
    uint32_t max_blocks = (uint32_t)ctx->n_cus * ctx->tune_blocks_per_cu;
    uint32_t blocks = (n + PTC_BLOCK - 1) / PTC_BLOCK;
    if (blocks > max_blocks) blocks = max_blocks;
    // the persistent kernel's static first blocks assume every block label (blockIdx % PTC_SHARDS) has the
    // same number of blocks; surplus blocks find nothing to do
    blocks = (blocks + PTC_SHARDS - 1u) / PTC_SHARDS * PTC_SHARDS;
    const uint32_t total_threads = blocks * PTC_BLOCK;
    // HBM spill columns: depth entries per thread (EXACT needs one frame per tree level too)
    // (WIDE pushes up to three siblings per level)
    const size_t depth = ctx->traversal == PT_TRAVERSE_WIDE ? 3u * (size_t)v.wide_depth + 2u : (size_t)v.max_depth + 2u;
    int rc = ptc_reserve(ctx, &ctx->spill, &ctx->spill_bytes, depth * total_threads * sizeof(uint2), "spill");
    if (rc != PT_OK) return rc;
    constexpr int S = 8;
    const float4* rays = reinterpret_cast<const float4*>(d_rays);
    uint4* out = reinterpret_cast<uint4*>(d_out);
    uint2* spill = reinterpret_cast<uint2*>(ctx->spill);
    dim3 grid(blocks), block(PTC_BLOCK);
#define PTC_LAUNCH(kern, ...) hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, v, rays, d_start, d_queue, d_count, n, out, spill, total_threads, ##__VA_ARGS__)
    if (ctx->traversal == PT_TRAVERSE_EXACT && ctx->tune_variant == 0) {
        if (d_queue) PTC_LAUNCH((k_trace_exact<S, true>));
        else PTC_LAUNCH((k_trace_exact<S, false>));
    } else if (ctx->tune_variant == 0 && ctx->traversal != PT_TRAVERSE_WIDE) {
        if (d_queue) PTC_LAUNCH((k_trace_fast<S, true>));
        else PTC_LAUNCH((k_trace_fast<S, false>));
    } else {
        uint32_t* const ticket = ctx->ticket + 512u * ctx->launch_parity;         // this launch's counters (256 B apart)
        uint32_t* const ticket_next = ctx->ticket + 512u * (ctx->launch_parity ^ 1u);
        ctx->launch_parity ^= 1u;
        const bool q = d_queue != nullptr;
#define PTC_PERS(SS, W) do { if (q) PTC_LAUNCH((k_trace_fast_persistent<SS, true, W>), ticket, ticket_next, refill, tsize); else PTC_LAUNCH((k_trace_fast_persistent<SS, false, W>), ticket, ticket_next, refill, tsize); } while (0)
        const uint32_t refill = ctx->tune_refill < 1u ? 1u : (ctx->tune_refill > 64u ? 64u : ctx->tune_refill);
        const uint32_t tsize = ctx->tune_ticket < 1u ? 1u : (ctx->tune_ticket > 1024u ? 1024u : ctx->tune_ticket);
        if (ctx->traversal == PT_TRAVERSE_EXACT) {
            if (q) PTC_LAUNCH((k_trace_exact_persistent<12, true>), ticket, ticket_next, refill, tsize);
            else PTC_LAUNCH((k_trace_exact_persistent<12, false>), ticket, ticket_next, refill, tsize);
        } else if (ctx->traversal == PT_TRAVERSE_WIDE) {
            switch (ctx->tune_ring) {
                case 8:  PTC_PERS(8, true); break;
                case 10: PTC_PERS(10, true); break;
                case 14: PTC_PERS(14, true); break;
                case 16: PTC_PERS(16, true); break;
                default: PTC_PERS(12, true); break;
            }
        } else switch (ctx->tune_ring) {
            case 8:  PTC_PERS(8, false); break;
            case 10: PTC_PERS(10, false); break;
            case 16: PTC_PERS(16, false); break;
            default: PTC_PERS(12, false); break;
        }
#undef PTC_PERS
    }
#undef PTC_LAUNCH
    return ptc_hip(ctx, hipGetLastError(), "trace kernel launch");
}
