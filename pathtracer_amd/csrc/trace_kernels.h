// trace_kernels.h — the device code of trace.hip (ring stack, ray tickets, the three persistent traversal kernels).
// A header of its own so that tests/cpp/emu_trace.cpp can compile the SAME source for the CPU (fibers for lanes, wave
// collectives resolved by a scheduler) and run it under the host's address sanitizer against the oracle; trace.hip is its
// only other includer.  See the head of trace.hip for what the kernels do.
#ifndef PTC_TRACE_KERNELS_H
#define PTC_TRACE_KERNELS_H

#define PTC_BLOCK 256
#ifndef PTC_POPS_PER_TRIP
#define PTC_POPS_PER_TRIP 1
#endif
#define PTC_FIRST 64u              /* rays of a wave's static first block (one per lane) */
#ifndef PTC_LEAF_EVERY
#define PTC_LEAF_EVERY 2           /* k_trace_scoped_persistent: the leaf test runs on every 2nd loop trip (1: every trip), see there */
#endif
// -DPTC_MARKERS (analysis builds only, tools/issue_mix.py): comment lines in the assembly at the start of the hot blocks of
// k_trace_scoped_persistent, so that the VALU opcode mix of each block can be read off `hipcc -S`
// (tests/cpp/emu_trace.cpp compiles this header for the CPU, where the occupancy attribute does not exist)
#ifdef PTC_EMU
#define PTC_WAVES_PER_EU(lo, hi)
#else
#define PTC_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
#ifdef PTC_MARKERS
#define PTC_MARK(name) asm volatile("; PTC_MARK " name)
#else
#define PTC_MARK(name) do { } while (0)
#endif

namespace {

// ---- per-lane ring stack: LDS window over an HBM column -----------------------------------------
// `ring` is the block's __shared__ array, slot-major ([slot][thread]); the column is
// [depth][thread] in HBM.  LDS and HBM accesses are kept in separate, differently-flavoured
// branches on purpose: if both sides were plain loads the compiler would select between the two
// ADDRESSES and emit one flat_load, which is far slower than ds_read_b64.
__device__ __forceinline__ uint32_t ptc_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Pointers with their address space spelled out: through generic pointers LLVM merges the LDS and the HBM arm of an access
// into one FLAT instruction on a selected address (slow, and not ordered with the wave's DS instructions).
typedef uint32_t ptc_u32x2 __attribute__((ext_vector_type(2)));      // (a plain vector: HIP's uint2 class has no address-space-qualified copies)
typedef __attribute__((address_space(3))) ptc_u32x2* ptc_lds_u2p;
typedef const __attribute__((address_space(3))) ptc_u32x2* ptc_lds_cu2p;
typedef __attribute__((address_space(1))) uint32_t* ptc_glb_u32p;
typedef const volatile __attribute__((address_space(1))) uint32_t* ptc_glb_vu32p;
__device__ __forceinline__ uint2 ptc_lds_get(ptc_lds_cu2p p) { const ptc_u32x2 v = *p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ void ptc_lds_put(ptc_lds_u2p p, uint32_t a, uint32_t b) { ptc_u32x2 v; v.x = a; v.y = b; *p = v; }

template <int S>
struct LaneStack {
    ptc_glb_u32p spill;  // &column[0][gtid] (as words: entry j at word 2 j spill_stride); depth stride = spill_stride
    uint32_t spill_stride;
    uint32_t idx;        // element index of the next free ring slot: slot * PTC_BLOCK + threadIdx.x
    uint32_t cnt;        // entries in LDS (the newest `cnt` of the stack)
    uint32_t base;       // entries below them, in HBM

    __device__ __forceinline__ void reset() { idx = threadIdx.x; cnt = 0u; base = 0u; }
    __device__ __forceinline__ bool empty() const { return (cnt | base) == 0u; }
    __device__ __forceinline__ uint32_t size() const { return cnt + base; }
    __device__ __forceinline__ void spill_store(uint32_t j, uint32_t a, uint32_t b) {
        __builtin_nontemporal_store(a, &spill[2u * (size_t)j * spill_stride]);
        __builtin_nontemporal_store(b, &spill[2u * (size_t)j * spill_stride + 1u]);
    }
    // volatile: keeps this a separate global load (otherwise LLVM selects between the LDS and HBM ADDRESSES and emits one
    // slow flat_load for every pop)
    __device__ __forceinline__ uint2 spill_load(uint32_t j) const {
        const ptc_glb_vu32p col = (ptc_glb_vu32p)&spill[2u * (size_t)j * spill_stride];
        uint2 e;
        e.x = col[0];
        e.y = col[1];
        return e;
    }
    __device__ __forceinline__ void push(uint2* __restrict__ ring_, uint32_t a, uint32_t b) {
        const ptc_lds_u2p ring = (ptc_lds_u2p)ring_;
        if (__builtin_expect(cnt == (uint32_t)S, 0)) {      // ring full: the next free slot holds the oldest entry
            const uint2 old = ptc_lds_get(ring + idx);
            spill_store(base, old.x, old.y);
            base += 1u;
            cnt -= 1u;
        }
        ptc_lds_put(ring + idx, a, b);
        idx += PTC_BLOCK;
        idx = ptc_umin(idx, idx - (uint32_t)(S * PTC_BLOCK));      // wraps: idx - S*BLOCK underflows unless idx >= S*BLOCK
        cnt += 1u;
    }
    __device__ __forceinline__ uint2 pop(const uint2* __restrict__ ring_) {     // caller checks !empty()
        const ptc_lds_cu2p ring = (ptc_lds_cu2p)ring_;
        // the LDS read is unconditional (its slot always exists) so that it stays a ds_read_b64;
        // the rare HBM refill overrides it
        const uint32_t prev = ptc_umin(idx - PTC_BLOCK, idx + (uint32_t)((S - 1) * PTC_BLOCK));
        uint2 e = ptc_lds_get(ring + prev);
        if (__builtin_expect(cnt == 0u, 0)) {
            base -= 1u;
            e = spill_load(base);
        } else {
            cnt -= 1u;
            idx = prev;
        }
        return e;
    }
};


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
    bool all_static;                 // the static first blocks cover the whole batch: no ticket word is ever touched

    __device__ __forceinline__ void init(uint32_t n) {
        blk_next = blk_end = 0u;
        exhausted = false;
        first_block = true;
        shard = blockIdx.x % PTC_SHARDS;
        shards_done = 0u;
        // shard length: a multiple of the static block, so that static and dynamic reservations tile a shard exactly
        shard_len = ((n + PTC_SHARDS * PTC_FIRST - 1u) / (PTC_SHARDS * PTC_FIRST)) * PTC_FIRST;
        waves_per_label = (gridDim.x / PTC_SHARDS) * (PTC_BLOCK / 64u);            // grid is a multiple of PTC_SHARDS
        // Small batches (the integrator's late iterations, the redo pass): when every shard is covered by the static blocks
        // a wave is done after its own - walking the shards' ticket words to find them all dry would cost the grid
        // PTC_SHARDS returning atomics per wave, ~50 000 on words that take ~90 per microsecond each.
        all_static = shard_len <= waves_per_label * PTC_FIRST;
        if (n == 0u) exhausted = true;
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
            } else if (all_static) {
                exhausted = true; blk_next = blk_end = 0u;
                break;
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

// ---- MIN_T: global-best pruning over the wide records -------------------------------------------------------------------
// The recursion never compares a box with a hit from ANOTHER subtree: it enters every box the ray meets, and skips the
// far child only if the near child's own result is not farther than the far box (`first.distance > far_min`, strict,
// bvh.rs:285).  This kernel prunes against the best hit of the whole ray, which in exact arithmetic loses nothing.  To stay
// close to the recursion where distances are EQUAL the comparison is strict only when the best hit was found after the
// entry was pushed - i.e. inside the near sibling's subtree, the one case in which the recursion compares at all - and
// non-strict otherwise.  "Found since the push" is exact: every entry on the stack when a hit is accepted is a pending
// sibling of an ancestor of the leaf, so `mark` = the stack height at the last accepted hit, lowered to the height after
// every pop, says it for each popped entry (height after the pop < mark).  Distances are non-negative, so their bit patterns
// order like the floats and "strict" is "one less": enter iff bits(best) - changed >= bits(entry).
//
// Persistent with dynamic refill: traversal lengths have a heavy tail (soup-1M primaries: the slowest of 64 neighbouring
// rays needs ~2.5x the mean), so a lane that finishes its ray takes the next one at once: each wave reserves blocks of
// `ticket_size` (64) consecutive rays from a ticket word and hands them to idle lanes by ballot + popcount rank whenever
// at least `refill` lanes are idle.  Every loop trip runs three predicated phases (record: slab tests + ordered push; leaf:
// triangle test; pop until something is still worth entering), so a lane can visit a record, test the leaf it descends
// into and pop its next subtree in ONE trip.
template <int S, bool QUEUED>
__global__ __launch_bounds__(PTC_BLOCK) void k_trace_mint_persistent(
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
    st.spill = (ptc_glb_u32p)(spill + gtid);
    st.spill_stride = total_threads;
    st.reset();
    if (sc.n_tris == 0u) {          // empty BVH: every ray misses (bvh.rs:364-368)
        for (uint32_t k = gtid; k < n; k += gridDim.x * PTC_BLOCK) out[QUEUED ? queue[k] : k] = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u);
        return;
    }

    RayTickets tk;
    tk.init(n);
    bool has_ray = false;
    DRay r = {};
    uint32_t ray_i = 0u, skip = PT_NO_TRIANGLE, cur = PTC_IDLE, best_id = PT_NO_TRIANGLE, mark = 0u;
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
                r = ptc_make_ray(ro, rays[2 * (size_t)ray_i + 1]);
                // QUEUED (the integrator's form): the start triangle rides in the ray's pad0
                skip = QUEUED ? __float_as_uint(ro.w) : (start ? start[ray_i] : PT_NO_TRIANGLE);
                best_t = pt_inf(); best_u = 0.0f; best_v = 0.0f; best_id = PT_NO_TRIANGLE;
                mark = 0u;
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
            // ---- phase 1: wide record (two levels of bvh.rs:259-325) ----
            if (!(cur & PTC_LEAF_BIT)) {
                const float4* p = sc.wide + (size_t)cur * 8u;
                const float4 lox = p[0], loy = p[1], loz = p[2], hix = p[3], hiy = p[4], hiz = p[5], rw = p[6];
                float k0, k1, k2, k3;
                uint32_t r0 = __float_as_uint(rw.x), r1 = __float_as_uint(rw.y), r2 = __float_as_uint(rw.z), r3 = __float_as_uint(rw.w);
                // the 24 subtractions and 24 products of the four slab tests, two boxes per packed instruction
                const PtcPair ax = ptc_slab_axis2(lox.x, lox.y, hix.x, hix.y, r.ox, r.ix), bx = ptc_slab_axis2(lox.z, lox.w, hix.z, hix.w, r.ox, r.ix);
                const PtcPair ay = ptc_slab_axis2(loy.x, loy.y, hiy.x, hiy.y, r.oy, r.iy), by = ptc_slab_axis2(loy.z, loy.w, hiy.z, hiy.w, r.oy, r.iy);
                const PtcPair az = ptc_slab_axis2(loz.x, loz.y, hiz.x, hiz.y, r.oz, r.iz), bz = ptc_slab_axis2(loz.z, loz.w, hiz.z, hiz.w, r.oz, r.iz);
                const bool h0 = ptc_slab_close(ax.t1.x, ax.t2.x, ay.t1.x, ay.t2.x, az.t1.x, az.t2.x, k0) && best_t >= k0;
                const bool h1 = ptc_slab_close(ax.t1.y, ax.t2.y, ay.t1.y, ay.t2.y, az.t1.y, az.t2.y, k1) && best_t >= k1;
                const bool h2 = ptc_slab_close(bx.t1.x, bx.t2.x, by.t1.x, by.t2.x, bz.t1.x, bz.t2.x, k2) && best_t >= k2;
                const bool h3 = ptc_slab_close(bx.t1.y, bx.t2.y, by.t1.y, by.t2.y, bz.t1.y, bz.t2.y, k3) && best_t >= k3;
                k0 = h0 ? k0 : pt_inf(); k1 = h1 ? k1 : pt_inf(); k2 = h2 ? k2 : pt_inf(); k3 = h3 ? k3 : pt_inf();
                // Ascending entry distance by odd-even transposition: six exchanges of NEIGHBOURS with a strict
                // compare, hence stable.  The record holds the children in REVERSE binary order (RR, RL, LR, LL), so
                // boxes with equal entry distance are entered right before left, as bvh.rs:266-270 does at every
                // binary node (`l_min < r_min` else right first).
#define PTC_CSWAP(ka, kb, ra, rb) { const bool sw = ka > kb; const float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; \
                                    const uint32_t tr = sw ? rb : ra; rb = sw ? ra : rb; ra = tr; }
                PTC_CSWAP(k0, k1, r0, r1) PTC_CSWAP(k2, k3, r2, r3) PTC_CSWAP(k1, k2, r1, r2)
                PTC_CSWAP(k0, k1, r0, r1) PTC_CSWAP(k2, k3, r2, r3) PTC_CSWAP(k1, k2, r1, r2)
#undef PTC_CSWAP
                if (k3 < pt_inf()) st.push(ring, r3, __float_as_uint(k3));
                if (k2 < pt_inf()) st.push(ring, r2, __float_as_uint(k2));
                if (k1 < pt_inf()) st.push(ring, r1, __float_as_uint(k1));
                cur = (k0 < pt_inf()) ? r0 : PTC_IDLE;
            }
            // ---- phase 2: leaf (bvh.rs:327-336) ----
            if ((cur & PTC_LEAF_BIT) && cur != PTC_IDLE) {
                const uint32_t tri = cur & PTC_INDEX_MASK;
                if (tri != skip) {
                    const float4* tp = sc.tris + (size_t)tri * PTC_TRI_STRIDE;
                    float d, u, v;
                    if (ptc_tri_test(r, tp[0], tp[1], tp[2], d, u, v) && d <= best_t) {
                        best_t = d; best_u = u; best_v = v; best_id = tri;
                        mark = st.size();
                    }
                }
                cur = PTC_IDLE;
            }
            // ---- phase 3: ONE pop per trip (see k_trace_scoped_persistent); an empty stack ends the ray ----
            if (cur == PTC_IDLE) {
                if (st.empty()) {
                    out[ray_i] = make_uint4(best_id, __float_as_uint(best_id == PT_NO_TRIANGLE ? 0.0f : best_t),
                                            __float_as_uint(best_u), __float_as_uint(best_v));
                    has_ray = false;
                } else {
                    const uint2 e = st.pop(ring);
                    const uint32_t height = st.size();
                    const uint32_t changed = height < mark ? 1u : 0u;      // the best hit lies in this entry's near sibling
                    mark = ptc_umin(mark, height);
                    if ((int32_t)(__float_as_uint(best_t) - changed) >= (int32_t)e.y) cur = e.x;
                }
            }
        }
    }
}

// ---- EXACT: the recursion's rule over the wide records (see the head of this file) -----------------------------------------
// redo_list / redo_count: rays this kernel does not answer (ptc_ray_is_regular false, or a triangle met at a non-finite
// distance); the launcher runs the REPLAY kernel over that list next.  Two counters alternate between calls like the
// ticket words: this launch clears the one the next call uses.
template <int S, bool QUEUED>
// (waves_per_eu 4..5: with the default target the register allocator spills the ray's inverse direction to scratch and
// reloads it at every record, and with a target of six waves it still spills 16 bytes per lane - scratch stores reach HBM;
// at five the kernel takes 76 VGPRs and no scratch, the descriptor admits five blocks per CU and the launcher asks for five)
__global__ __launch_bounds__(PTC_BLOCK) PTC_WAVES_PER_EU(4, 5) void k_trace_scoped_persistent(
    DevSceneView sc, const float4* __restrict__ rays, const uint32_t* __restrict__ start,
    const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t n,
    uint4* __restrict__ out, uint2* __restrict__ spill, uint32_t total_threads, uint32_t* __restrict__ ticket,
    uint32_t* __restrict__ ticket_next, uint32_t refill, uint32_t ticket_size, uint32_t* __restrict__ redo_list,
    uint32_t* __restrict__ redo_count, uint32_t* __restrict__ redo_count_next) {
    __shared__ uint2 ring[S * PTC_BLOCK];
    const uint32_t gtid = blockIdx.x * PTC_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (count_ptr) n = *count_ptr;
    if (gtid < PTC_SHARDS) ticket_next[gtid * 64u] = 0u;      // see k_trace_mint_persistent
    if (gtid == 0u) *redo_count_next = 0u;
    LaneStack<S> st;
    st.spill = (ptc_glb_u32p)(spill + gtid);
    st.spill_stride = total_threads;
    st.reset();
    if (sc.n_tris == 0u) {          // empty BVH: every ray misses (bvh.rs:364-368)
        for (uint32_t k = gtid; k < n; k += gridDim.x * PTC_BLOCK) out[QUEUED ? queue[k] : k] = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u);
        return;
    }

    RayTickets tk;
    tk.init(n);
    bool has_ray = false;
    // the ray as nine scalars (a DRay is made of them where the step functions want one): as a struct its inverse direction
    // stayed in memory, and LLVM moved that to LDS (3 KB per block, a read per record)
    float rox = 0.0f, roy = 0.0f, roz = 0.0f, rdx = 0.0f, rdy = 0.0f, rdz = 0.0f, rix = 0.0f, riy = 0.0f, riz = 0.0f;
    uint32_t ray_i = 0u, skip = PT_NO_TRIANGLE, cur = PTC_IDLE, best_id = PT_NO_TRIANGLE;
    float best_t = pt_inf(), best_u = 0.0f, best_v = 0.0f;
    float m_cur = pt_inf();          // minimum t found since the newest stack entry was pushed (inf: none)
#if PTC_LEAF_EVERY > 1
    uint32_t leaf_trip = 0u;         // loop trips of this wave (wave-uniform): the leaf test runs on every PTC_LEAF_EVERY-th
#endif
    bool sign_ok = true;             // this lane's ray allows the sign-select wide step (an idle lane does not object)
    uint32_t near_x = 0u, near_y = 16u, near_z = 32u;      // byte offsets of its near planes inside a wide record

    for (;;) {
        // ---- refill ----
        const unsigned long long idle = __ballot(!has_ray);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (!tk.exhausted && (n_idle >= refill || n_idle == 64u)) {
            uint32_t k;
            if (tk.take(idle, n_idle, lane, !has_ray, n, ticket, ticket_size, k)) {
                ray_i = QUEUED ? queue[k] : k;
                const float4 ro = rays[2 * (size_t)ray_i];
                const DRay r = ptc_make_ray(ro, rays[2 * (size_t)ray_i + 1]);
                skip = QUEUED ? __float_as_uint(ro.w) : (start ? start[ray_i] : PT_NO_TRIANGLE);
                if (ptc_ray_is_regular(r)) {
                    best_t = pt_inf(); best_u = 0.0f; best_v = 0.0f; best_id = PT_NO_TRIANGLE;
                    m_cur = pt_inf();
                    rox = r.ox; roy = r.oy; roz = r.oz; rdx = r.dx; rdy = r.dy; rdz = r.dz; rix = r.ix; riy = r.iy; riz = r.iz;
                    sign_ok = ptc_ray_sign_select(r);
                    near_x = r.ix < 0.0f ? 48u : 0u; near_y = r.iy < 0.0f ? 64u : 16u; near_z = r.iz < 0.0f ? 80u : 32u;
                    st.reset();
                    cur = sc.root_ref;
                    has_ray = true;
                } else {
                    redo_list[atomicAdd(redo_count, 1u)] = ray_i;
                }
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (tk.exhausted) break;
            continue;
        }
        // The three blocks below are executed by the wave whenever ANY lane needs them, so what each costs is set by how many
        // lanes share an execution (tools/wave_sim.py replays the model's trip logs through this loop: the record step runs
        // with ~54 of 64 lanes, the leaf test with ~12, and a pop loop iterated until the LAST lane had found an entry ran
        // with ~8, four iterations per trip).  So a trip pops ONCE: a lane whose entry was not entered (or was a RESTORE)
        // pops again on the next trip, beside the others' next record, instead of holding them up (EXACT 1 727 -> 1 929 Mrays/s, bounce
        // rays 1 485 -> 1 675; two pops per trip - PTC_POPS_PER_TRIP - or three, and making lanes wait at a leaf until 8-24
        // are there, were slower or equal; so was touching the next record's line one trip early).
        const bool at_record = has_ray && !(cur & PTC_LEAF_BIT);
#ifdef PTC_EMU
        // (the CPU run of this source - tests/cpp/emu_trace.cpp - needs its collectives in wave-uniform control flow: there the
        // vote is taken by every lane, outside the branch - the same mask as the vote of the lanes inside it, which is what the
        // GPU build takes: measured 1.2 % faster than this form)
        const bool wave_sign_select = sc.sign_select_ok && __ballot(at_record && !sign_ok) == 0ull;
#define PTC_WAVE_SIGN_SELECT wave_sign_select
#else
#define PTC_WAVE_SIGN_SELECT (sc.sign_select_ok && __ballot(!sign_ok) == 0ull)
#endif
        if (at_record) {
            // ---- phase 1: wide record = bvh.rs:259-325 at a binary node and at the child it enters ----
            {
                PtcWideStep s;
                const DRay r = {rox, roy, roz, rdx, rdy, rdz, rix, riy, riz};
                // Sign-select form (trace_step.h) when every ray of the wave has a finite, non-zero inverse direction and the
                // scene allows it: each lane loads the near planes (lo or hi by the sign of its inverse direction) and the far
                // planes, and the 24 min / max that pick them drop out.  Uniform base + 32-bit lane offset addressing.
                if (PTC_WAVE_SIGN_SELECT) {
                    PTC_MARK("record_sign");
                    const char* base = reinterpret_cast<const char*>(sc.wide);
                    const uint32_t rec = cur * 128u;
                    const float4 nearx = *reinterpret_cast<const float4*>(base + (rec + near_x));
                    const float4 neary = *reinterpret_cast<const float4*>(base + (rec + near_y));
                    const float4 nearz = *reinterpret_cast<const float4*>(base + (rec + near_z));
                    const float4 farx = *reinterpret_cast<const float4*>(base + (rec + (48u - near_x)));
                    const float4 fary = *reinterpret_cast<const float4*>(base + (rec + (80u - near_y)));
                    const float4 farz = *reinterpret_cast<const float4*>(base + (rec + (112u - near_z)));
                    const float4 rw = *reinterpret_cast<const float4*>(base + (rec + 96u));
                    ptc_wide_step_sign(r, nearx, neary, nearz, farx, fary, farz, rw, s);
                } else {
                    PTC_MARK("record_general");
                    const float4* p = sc.wide + (size_t)cur * 8u;
                    const float4 lox = p[0], loy = p[1], loz = p[2], hix = p[3], hiy = p[4], hiz = p[5], rw = p[6];
                    ptc_wide_step(r, lox, loy, loz, hix, hiy, hiz, rw, s);
                }
                PTC_MARK("pushes");
                // a push opens a scope: a finite m_cur is parked below it (one RESTORE covers all pushes of this visit)
                if ((s.v0 || s.v1 || s.v2) && m_cur < pt_inf()) { st.push(ring, PTC_RESTORE, __float_as_uint(m_cur)); m_cur = pt_inf(); }
                if (s.v0) st.push(ring, s.ref0, __float_as_uint(s.e0));
                if (s.v1) st.push(ring, s.ref1, __float_as_uint(s.e1));
                if (s.v2) st.push(ring, s.ref2, __float_as_uint(s.e2));
                cur = s.cur;
            }
        }
        // ---- phase 2: leaf (bvh.rs:327-336) ----
        // The leaf test is executed by the wave whenever a lane needs it - with ~12 of 64 lanes on nearly every trip, a third of the
        // kernel's VALU instructions.  So it runs on every SECOND trip only (PTC_LEAF_EVERY), or at once when no lane of the wave is
        // at a record: a lane that reaches a leaf on an odd trip keeps its leaf for one trip, and the test then serves about twice
        // the lanes.  The wait is bounded by one trip, which is what separates this from the forms that lost (lanes waiting until 8 - 24
        // of them are there; every third trip: equal): primaries 0.997 -> 0.973 ms, bounce rays 1.173 -> 1.149, config 3 230.4 -> 236.5
        // spp/s, same bits (a lane's own sequence of visits does not change).  Also tried on top, all equal to the plain form:
        // every trip once the tickets are gone, and at once when 16 or 24 lanes wait.
#if PTC_LEAF_EVERY > 1
        leaf_trip += 1u;
        const bool leaf_now = (leaf_trip % PTC_LEAF_EVERY) == 0u || __ballot(has_ray && !(cur & PTC_LEAF_BIT)) == 0ull;
#else
        const bool leaf_now = true;
#endif
        if (leaf_now) {
            if (has_ray && (cur & PTC_LEAF_BIT) && cur != PTC_IDLE) {
                const uint32_t tri = cur & PTC_INDEX_MASK;
                if (tri != skip) {
                    PTC_MARK("leaf");
                    const float4* tp = sc.tris + (size_t)tri * PTC_TRI_STRIDE;
                    float d, u, v;
                    const DRay r = {rox, roy, roz, rdx, rdy, rdz, rix, riy, riz};
                    if (ptc_tri_test(r, tp[0], tp[1], tp[2], d, u, v)) {
                        if (!(pt_fabs(d) < pt_inf())) {          // Some(inf) / Some(NaN): only the frames get those right
                            redo_list[atomicAdd(redo_count, 1u)] = ray_i;
                            has_ray = false;
                            sign_ok = true;
                        }
                        if (d <= best_t) { best_t = d; best_u = u; best_v = v; best_id = tri; }    // later visit wins a tie (bvh.rs:299-303)
                        m_cur = d < m_cur ? d : m_cur;
                    }
                }
                cur = PTC_IDLE;
            }
        }
        // ---- phase 3: pop; an empty stack ends the ray ----
#pragma unroll
        for (int rep = 0; rep < PTC_POPS_PER_TRIP; ++rep) {
            if (has_ray && cur == PTC_IDLE) {
                if (st.empty()) {
                    out[ray_i] = make_uint4(best_id, __float_as_uint(best_id == PT_NO_TRIANGLE ? 0.0f : best_t),
                                            __float_as_uint(best_u), __float_as_uint(best_v));
                    has_ray = false;
                    sign_ok = true;
                } else {
                    PTC_MARK("pop");
                    const uint2 e = st.pop(ring);
                    const float ey = __uint_as_float(e.y);
                    const bool restore = e.x == PTC_RESTORE;                // {RESTORE, the enclosing scope's minimum so far}
                    const bool visit = !restore && m_cur > ey;             // first_distance > second_child_min (bvh.rs:285)
                    if (restore) m_cur = ey < m_cur ? ey : m_cur;
                    if (!restore && (e.x & PTC_COMPANION)) {                // the far child's other grandchild lies directly below
                        const uint2 c = st.pop(ring);
                        if (visit) {                                        // entering the far child: its far grandchild waits for the near one
                            if (m_cur < pt_inf()) { st.push(ring, PTC_RESTORE, __float_as_uint(m_cur)); m_cur = pt_inf(); }
                            st.push(ring, c.x, c.y);
                        }
                    }
                    if (visit) cur = e.x & ~PTC_COMPANION;
                }
            }
        }
        PTC_MARK("loop_end");
    }
}

// ---- REPLAY: the recursion frame by frame over the binary nodes -------------------------------------------------------------
// Leaf refs carry PTC_LEAF_BIT with PTC_FRAME_BIT clear (indices < 2^30), internal refs have both clear, so an A frame
// never looks like a B frame.  QUEUED: ray indices come from `queue`; PADSTART: the start triangle rides in the ray's pad0
// (the integrator's rays) instead of `start`.
template <int S, bool QUEUED, bool PADSTART>
__global__ __launch_bounds__(PTC_BLOCK) void k_trace_replay_persistent(
    DevSceneView sc, const float4* __restrict__ rays, const uint32_t* __restrict__ start,
    const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t n,
    uint4* __restrict__ out, uint2* __restrict__ spill, uint32_t total_threads, uint32_t* __restrict__ ticket,
    uint32_t* __restrict__ ticket_next, uint32_t refill, uint32_t ticket_size) {
    __shared__ uint2 ring[S * PTC_BLOCK];
    const uint32_t gtid = blockIdx.x * PTC_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (count_ptr) n = *count_ptr;
    if (gtid < PTC_SHARDS) ticket_next[gtid * 64u] = 0u;      // see k_trace_mint_persistent
    LaneStack<S> st;
    st.spill = (ptc_glb_u32p)(spill + gtid);
    st.spill_stride = total_threads;
    st.reset();
    if (sc.n_tris == 0u) {          // empty BVH: every ray misses (bvh.rs:364-368)
        for (uint32_t k = gtid; k < n; k += gridDim.x * PTC_BLOCK) out[QUEUED ? queue[k] : k] = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u);
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
                r = ptc_make_ray(ro, rays[2 * (size_t)ray_i + 1]);
                skip = PADSTART ? __float_as_uint(ro.w) : (start ? start[ray_i] : PT_NO_TRIANGLE);
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
                const float4* q = sc.nodes + (size_t)cur * 4u;
                const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                float lt, rt;
                const bool lh = ptc_slab(r, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, lt);    // bvh.rs:260
                const bool rh = ptc_slab(r, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, rt);    // bvh.rs:261
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
                    const float4* tp = sc.tris + (size_t)tri * PTC_TRI_STRIDE;
                    float d, u, v;
                    if (ptc_tri_test(r, tp[0], tp[1], tp[2], d, u, v)) { ret_id = tri; ret_t = d; }
                }
                descending = false;
            }
            // ---- phase 3: return through ONE frame per trip (a lane still returning goes on next trip, beside the others' next
            // node, instead of holding the wave in this block: see k_trace_scoped_persistent) ----
            if (!descending) {
                if (st.empty()) {
                    uint4 o = make_uint4(PT_NO_TRIANGLE, 0u, 0u, 0u);
                    if (ret_id != PTC_NONE_ID) {
                        // u, v of the winner: the same arithmetic on the same inputs gives the same bits
                        const float4* tp = sc.tris + (size_t)ret_id * PTC_TRI_STRIDE;
                        float d, u, v;
                        ptc_tri_test(r, tp[0], tp[1], tp[2], d, u, v);
                        o = make_uint4(ret_id, __float_as_uint(d), __float_as_uint(u), __float_as_uint(v));
                    }
                    out[ray_i] = o;
                    has_ray = false;
                } else {
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
}

}  // namespace

#endif
