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
// column in HBM ([depth][thread], coalesced) and comes back when the ring runs dry.  Wide records are
// fetched as 7 x 16-byte loads, binary nodes as 4, triangles as 3 x 16-byte loads per lane.
//
// Three flavours (pt_set_traversal), all persistent kernels that hand the next ray to a lane as soon as it is done
// (RayTickets):
//  EXACT  (default) returns what traverse_recursive returns, on every ray, by construction - and walks the 128-byte
//         records of device_types.h, two levels of the binary tree per fetch.  The recursion's rule is LOCAL: the far
//         child of a node is entered iff the near child's own result is farther than the far box (bvh.rs:285); there is
//         no pruning against hits from elsewhere, and the result is the minimum-t hit among the leaves so visited, the
//         later visit winning a tie (bvh.rs:299-303).  So the answer needs one id (the best hit of the whole ray, replaced
//         iff t <= best, in the recursion's visiting order) and, for every pending far sibling, the minimum t found since
//         it was pushed: the "scope minimum".  Scopes nest like the stack, so ONE register holds the innermost scope's
//         minimum (m_cur): a push starts a new scope (m_cur = inf, after parking a finite m_cur in a RESTORE entry), a
//         pop closes one (the decision `m_cur > entry`, then the parked value folds back in: min).  Per loop trip a lane may
//         visit a record, test a leaf (that phase runs on every second trip of the wave, trace_kernels.h) and pop once.  The per-record
//         decisions are ptc_wide_step (trace_step.h), shared with the host-side model that the CPU tests check against
//         the oracle.  A ray with a non-finite component, or one that meets a triangle at a non-finite distance (where
//         Some(inf) / Some(NaN) and None part ways in bvh.rs:280-303), is handed to REPLAY through a redo list.
//  REPLAY replays the recursion frame by frame over the binary nodes: an "A" frame {far ref, far tmin} waits for the near
//         result, a "B" frame {first id | tag, first distance} waits for the far result.  The reference implementation
//         of the rule on the device, the redo path of EXACT, and selectable on its own.
//  MIN_T  the fastest: global-best pruning over the wide records (enter a box iff best >= its entry distance, the four
//         boxes of a record nearest first).  Returns a minimum-t hit; which triangle of an exact equal-t tie, or of two
//         hits an ulp apart where a box's entry distance rounds above its own triangle's t, is not the recursion's
//         choice in general.
//
// All float arithmetic is written in the reference's operation order and the file is compiled
// with -ffp-contract=off: no FMA, IEEE division, minNum/maxNum for f32::min/max.

#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_types.h"
#include "trace_step.h"

#include "trace_kernels.h"

// Everything a launch of one traversal flavour needs beyond its arguments: grid, spill columns, ticket words.
// `scratch` selects where the spill columns, ticket words and redo list live: the context's own buffers (grown on demand),
// the fixed array of trace_noalloc.hip (nothing is allocated then), or a slot of pt_trace_closest_dev_submit (its own
// stream and grown-on-demand buffers, so that batches on different slots can overlap).
int ptc_launch_trace_in(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays, const uint32_t* d_start,
                        const uint32_t* d_queue, const uint32_t* d_count, uint32_t n, pt_hit* d_out, ptc_trace_scratch* scratch) {
    if (n == 0) return PT_OK;
    if (n > 0xF0000000u) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "more than 0xF0000000 rays in one batch");
    const DevSceneView& v = scene->view;
    // EXACT re-derives the boxes of an odd-depth node's children from their own children's boxes; on a caller's tree whose
    // stored boxes are not those joins (DevSceneView::joins_exact) the frame replay, which reads the stored boxes, answers
    const uint32_t mode = (ctx->traversal == PT_TRAVERSE_EXACT && !v.joins_exact) ? PT_TRAVERSE_REPLAY : ctx->traversal;
    // resident grid: waves_per_simd blocks of 256 per CU at most (a persistent kernel must be resident as a whole: a block
    // that starts only when another ends finds the tickets gone); fewer when the batch is small
    uint32_t per_cu = ctx->tune_blocks_per_cu;
    if (mode == PT_TRAVERSE_EXACT) {
        // k_trace_scoped_persistent is built for at most five waves per SIMD (its waves_per_eu attribute), and the dispatcher
        // holds it to that: a sixth block per CU would only start when another ends (measured: bounce rays 1.46 -> 1.61 ms);
        // the LDS ring caps it as well
        const uint32_t ring = (ctx->tune_ring == 8u || ctx->tune_ring == 10u || ctx->tune_ring == 16u) ? ctx->tune_ring : 12u;
        per_cu = std::min<uint32_t>(std::min<uint32_t>(per_cu, 5u), (160u * 1024u) / (ring * PTC_BLOCK * (uint32_t)sizeof(uint2)));
    }
    uint32_t max_blocks = (uint32_t)ctx->n_cus * per_cu;
    if (scratch && scratch->max_blocks && max_blocks > scratch->max_blocks) max_blocks = scratch->max_blocks;
    uint32_t blocks = (n + PTC_BLOCK - 1) / PTC_BLOCK;
    if (blocks > max_blocks) blocks = max_blocks;
    // the persistent kernels' static first blocks assume every block label (blockIdx % PTC_SHARDS) has the
    // same number of blocks; surplus blocks find nothing to do
    blocks = (blocks + PTC_SHARDS - 1u) / PTC_SHARDS * PTC_SHARDS;
    const uint32_t total_threads = blocks * PTC_BLOCK;
    const size_t depth = ptc_spill_depth(v, mode);          // HBM spill columns, entries per thread
    const size_t spill_need = depth * total_threads * sizeof(uint2);
    uint2* spill = nullptr;
    uint32_t *tickets = nullptr, *redo_list = nullptr;
    hipStream_t stream = (scratch && scratch->stream) ? scratch->stream : ctx->stream;
    if (scratch) {
        const size_t redo_need = mode == PT_TRAVERSE_EXACT ? (size_t)n * sizeof(uint32_t) : 0u;
        if (scratch->grow && (spill_need > scratch->spill_bytes || redo_need > scratch->redo_bytes)) {
            // a slot's own buffers (pt_trace_closest_dev_submit): what is in flight on its stream still uses the old ones
            int rc = ptc_hip(ctx, hipStreamSynchronize(stream), "hipStreamSynchronize (slot)");
            if (rc == PT_OK) rc = ptc_reserve(ctx, &scratch->spill, &scratch->spill_bytes, spill_need, "spill (slot)");
            if (rc == PT_OK) rc = ptc_reserve(ctx, reinterpret_cast<void**>(&scratch->redo), &scratch->redo_bytes, redo_need, "redo list (slot)");
            if (rc != PT_OK) return rc;
        }
        if (spill_need > scratch->spill_bytes) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "scene too deep for the fixed traversal scratch");
        if (redo_need > scratch->redo_bytes) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "batch too large for the fixed traversal scratch");
        spill = reinterpret_cast<uint2*>(scratch->spill);
        tickets = scratch->tickets;
        redo_list = scratch->redo;
    } else {
        int rc = ptc_reserve(ctx, &ctx->spill, &ctx->spill_bytes, spill_need, "spill");
        if (rc != PT_OK) return rc;
        if (mode == PT_TRAVERSE_EXACT && (rc = ptc_reserve(ctx, &ctx->redo, &ctx->redo_bytes, (size_t)n * sizeof(uint32_t), "redo list")) != PT_OK) return rc;
        spill = reinterpret_cast<uint2*>(ctx->spill);
        tickets = ctx->ticket;
        redo_list = reinterpret_cast<uint32_t*>(ctx->redo);
    }
    uint32_t& parity = scratch ? *scratch->launch_parity : ctx->launch_parity;
    uint32_t& redo_parity = scratch ? *scratch->redo_parity : ctx->redo_parity;
    const float4* rays = reinterpret_cast<const float4*>(d_rays);
    uint4* out = reinterpret_cast<uint4*>(d_out);
    dim3 grid(blocks), block(PTC_BLOCK);
    const bool q = d_queue != nullptr;
    // the integrator's launches (a queue of path positions: tens of millions of incoherent rays whose queue neighbours are samples of
    // the same pixel) refill in larger groups than a caller's batch: config 3 223 -> 230 spp/s, config 4 54.7 -> 56.1, where 2 M-ray
    // batches of bounce rays prefer 16 (tools/sweep_knobs_render.sh)
    const uint32_t refill_want = q ? ctx->tune_refill_queued : ctx->tune_refill;
    const uint32_t refill = refill_want < 1u ? 1u : (refill_want > 64u ? 64u : refill_want);
    const uint32_t tsize = ctx->tune_ticket < 1u ? 1u : (ctx->tune_ticket > 1024u ? 1024u : ctx->tune_ticket);
    // this launch's ticket words (sets 2 KiB apart) and the set it clears for the next launch
    auto next_tickets = [&](uint32_t*& cur_set, uint32_t*& next_set) {
        cur_set = tickets + 512u * parity;
        next_set = tickets + 512u * (parity ^ 1u);
        parity ^= 1u;
    };
    uint32_t *tk = nullptr, *tk_next = nullptr;
    next_tickets(tk, tk_next);
#define PTC_LAUNCH(kern, qptr, cptr, ...) hipLaunchKernelGGL(kern, grid, block, 0, stream, v, rays, d_start, qptr, cptr, n, out, spill, total_threads, tk, tk_next, refill, tsize, ##__VA_ARGS__)
    if (mode == PT_TRAVERSE_REPLAY) {
        if (q) PTC_LAUNCH((k_trace_replay_persistent<12, true, true>), d_queue, d_count);
        else PTC_LAUNCH((k_trace_replay_persistent<12, false, false>), d_queue, d_count);
    } else if (mode == PT_TRAVERSE_MIN_T) {
        if (q) PTC_LAUNCH((k_trace_mint_persistent<12, true>), d_queue, d_count);
        else PTC_LAUNCH((k_trace_mint_persistent<12, false>), d_queue, d_count);
    } else {
        uint32_t* const redo_count = tickets + 1024u + 64u * redo_parity;
        uint32_t* const redo_count_next = tickets + 1024u + 64u * (redo_parity ^ 1u);
        redo_parity ^= 1u;
#define PTC_SCOPED(SS) do { if (q) PTC_LAUNCH((k_trace_scoped_persistent<SS, true>), d_queue, d_count, redo_list, redo_count, redo_count_next); \
                            else PTC_LAUNCH((k_trace_scoped_persistent<SS, false>), d_queue, d_count, redo_list, redo_count, redo_count_next); } while (0)
        switch (ctx->tune_ring) {             // PTC_RING
            case 8:  PTC_SCOPED(8); break;
            case 10: PTC_SCOPED(10); break;
            case 16: PTC_SCOPED(16); break;
            default: PTC_SCOPED(12); break;
        }
#undef PTC_SCOPED
        // the redo pass: whatever the scoped kernel listed - usually nothing, and the launch finds a zero count and ends; its
        // grid is a quarter of the main one (an empty full-size launch costs ~10 us, and the list is short when it is not
        // empty; the spill columns are sized for the larger grid)
        next_tickets(tk, tk_next);
        grid = dim3(std::max(PTC_SHARDS, blocks / 4u / PTC_SHARDS * PTC_SHARDS));
        if (q) PTC_LAUNCH((k_trace_replay_persistent<12, true, true>), redo_list, redo_count);
        else PTC_LAUNCH((k_trace_replay_persistent<12, true, false>), redo_list, redo_count);
    }
#undef PTC_LAUNCH
    return ptc_hip(ctx, hipGetLastError(), "trace kernel launch");
}

int ptc_launch_trace(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays, const uint32_t* d_start,
                     const uint32_t* d_queue, const uint32_t* d_count, uint32_t n, pt_hit* d_out) {
    return ptc_launch_trace_in(ctx, scene, d_rays, d_start, d_queue, d_count, n, d_out, nullptr);
}
