// trace_noalloc.hip — pt_trace_closest_dev_noalloc: the closest-hit entry point for callers that cannot allocate
// (tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  It is a BLOCKING call like
// pt_trace_closest_dev (the library's own stream, events around the kernels, hipEventSynchronize before it returns), so it
// cannot be captured into a HIP graph: what it removes is the allocation, not the synchronisation.
//
// pt_trace_closest_dev keeps three scratch buffers in the context and grows them with hipMalloc on demand: the HBM columns
// the per-lane traversal stacks spill to, the ticket words of the persistent kernels, and the redo list of
// PT_TRAVERSE_EXACT.  This variant runs the SAME kernels (trace.hip) with those three carved out of one file-scope device
// array of fixed size instead, so the call allocates nothing: no hipMalloc, no hipFree, no growing workspace.
//
//   * every GPU the library runs on holds its own copy of the array (PTC_NOALLOC_BYTES, about 144 MiB) from the moment the
//     code object is loaded until the process ends;
//   * the array's name is a host-side stand-in, not a device address: its address on a device comes from
//     hipGetSymbolAddress, once per device with that device current (ptc_noalloc_setup, called by pt_init), and only that
//     pointer is ever handed to kernels, hipMemsetAsync or anything else;
//   * all contexts on one GPU share that GPU's array, and it keeps the last call's contents: every call clears the ticket
//     words it depends on first, and calls that overlap in time on one GPU race (one call at a time per GPU);
//   * limits, rejected with PT_ERR_UNSUPPORTED and a message: at most PT_NOALLOC_MAX_RAYS rays per call; the scene's tree
//     must be shallow enough for at least PTC_SHARDS (8) blocks' worth of stack columns to fit (the grid is cut down to what fits, so
//     deep scenes run with fewer resident waves than pt_trace_closest_dev would use).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>

#include "device_types.h"

#define PTC_NOALLOC_SPILL_BYTES (128u << 20)
#define PTC_NOALLOC_REDO_BYTES  ((size_t)PT_NOALLOC_MAX_RAYS * sizeof(uint32_t))
#define PTC_NOALLOC_BYTES       ((size_t)PTC_NOALLOC_SPILL_BYTES + PTC_NOALLOC_REDO_BYTES + PTC_TICKET_BYTES)
#define PTC_NOALLOC_MAX_DEVICES 64

static __device__ __align__(256) unsigned char scratch[PTC_NOALLOC_BYTES];

namespace {
struct DeviceScratch {
    unsigned char* ptr = nullptr;        // device address of THIS device's copy of `scratch`
    uint32_t launch_parity = 0, redo_parity = 0;
};
DeviceScratch g_dev[PTC_NOALLOC_MAX_DEVICES];
std::mutex g_mu;
}  // namespace

// Called by pt_init with ctx->device current.  Resolves this device's address of the array (once per device).
int ptc_noalloc_setup(pt_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= PTC_NOALLOC_MAX_DEVICES) return PT_OK;      // no fixed scratch there: the entry point says so
    std::lock_guard<std::mutex> g(g_mu);
    DeviceScratch& d = g_dev[ctx->device];
    if (!d.ptr) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return ptc_hip(ctx, e == hipSuccess ? hipErrorInvalidSymbol : e, "hipGetSymbolAddress(scratch)");
        d.ptr = static_cast<unsigned char*>(p);
    }
    return PT_OK;
}

static int pt_trace_closest_dev_noalloc_body(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                                            const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!scene || (n && (!d_rays || !d_out))) return ptc_fail(ctx, PT_ERR_INVALID, "pt_trace_closest_dev_noalloc: NULL argument");
    if (n > PT_NOALLOC_MAX_RAYS)
        return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_trace_closest_dev_noalloc: more than PT_NOALLOC_MAX_RAYS (4194304) rays in one call");
    if (ctx->device < 0 || ctx->device >= PTC_NOALLOC_MAX_DEVICES || !g_dev[ctx->device].ptr)
        return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_trace_closest_dev_noalloc: no fixed scratch on this device");
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    ctx->stats = pt_stats{};
    if (n == 0) return PT_OK;
    DeviceScratch& d = g_dev[ctx->device];
    // the stack columns need `depth` 8-byte entries per resident thread: cut the grid down to what the fixed area holds
    const DevSceneView& v = scene->view;
    const uint32_t mode = (ctx->traversal == PT_TRAVERSE_EXACT && !v.joins_exact) ? PT_TRAVERSE_REPLAY : ctx->traversal;   // as ptc_launch_trace_in
    const size_t depth = ptc_spill_depth(v, mode);
    const size_t per_block = depth * 256u * sizeof(uint2);
    const size_t fit = (PTC_NOALLOC_SPILL_BYTES / per_block) / PTC_SHARDS * PTC_SHARDS;      // the grid is a multiple of PTC_SHARDS
    if (fit < PTC_SHARDS) return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_trace_closest_dev_noalloc: the scene's BVH is too deep for the fixed traversal scratch");
    ptc_trace_scratch s{};
    s.spill = d.ptr;
    s.spill_bytes = PTC_NOALLOC_SPILL_BYTES;
    s.redo = reinterpret_cast<uint32_t*>(d.ptr + PTC_NOALLOC_SPILL_BYTES);
    s.redo_bytes = PTC_NOALLOC_REDO_BYTES;
    s.tickets = reinterpret_cast<uint32_t*>(d.ptr + PTC_NOALLOC_SPILL_BYTES + PTC_NOALLOC_REDO_BYTES);
    s.max_blocks = (uint32_t)std::min<size_t>(fit, 1u << 20);
    s.launch_parity = &d.launch_parity;
    s.redo_parity = &d.redo_parity;
    // the array keeps the last call's contents (and another context's): the ticket words and redo counters start at zero
    d.launch_parity = 0u;
    d.redo_parity = 0u;
    PTC_HIP(ctx, hipMemsetAsync(s.tickets, 0, PTC_TICKET_BYTES, ctx->stream));
    PTC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    const int rc = ptc_launch_trace_in(ctx, scene, d_rays, d_start_tri, nullptr, nullptr, n, d_out, &s);
    if (rc != PT_OK) return rc;
    PTC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    PTC_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    float ms = 0.0f;
    PTC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->stats.trace_ms = ms; ctx->stats.total_ms = ms;
    ctx->stats.trace_rays = n; ctx->stats.trace_launches = 1;
    return PT_OK;
}

extern "C" int pt_trace_closest_dev_noalloc(pt_ctx* ctx, const pt_scene* scene, const pt_ray* d_rays,
                                            const uint32_t* d_start_tri, uint32_t n, pt_hit* d_out) {
    return ptc_guard(ctx, "pt_trace_closest_dev_noalloc", [&] { return pt_trace_closest_dev_noalloc_body(ctx, scene, d_rays, d_start_tri, n, d_out); });
}
