// emu_stubs.cpp — what the CPU build of the library (libptcore_emu.so, see emu_rt.h) leaves out: lbvh.hip (the device-side BVH
// build, node flattening and wide collapse; it sorts with hipCUB).  pt_scene_create's host-side flattening and collapse stand in
// for the latter two (PTC_FLATTEN=1 PTC_WIDE_COLLAPSE=1, which tests/test_emu_lib.py sets; on the GPU the suite holds the two
// paths byte-identical), PT_SCENE_BUILD_LBVH is refused.  TEST INFRASTRUCTURE.
#include "../../pathtracer_amd/csrc/device_types.h"

static int refuse(pt_ctx* ctx, const char* what) {
    return ptc_fail(ctx, PT_ERR_UNSUPPORTED, std::string(what) + " is not part of the CPU emulation build (set PTC_FLATTEN=1 PTC_WIDE_COLLAPSE=1)");
}
int ptc_build_lbvh(pt_ctx* ctx, const float4*, uint32_t, float4**, float4**, uint32_t*, uint32_t*, uint32_t*, float*) { return refuse(ctx, "the device-side BVH build"); }
int ptc_flatten_nodes(pt_ctx* ctx, const pt_bvh_node*, uint32_t, uint32_t, float4**, uint32_t*, bool*) { return refuse(ctx, "device-side node flattening"); }
int ptc_build_wide_from_nodes(pt_ctx* ctx, const float4*, uint32_t, float4**, uint32_t*, uint32_t*, uint32_t*, bool*) { return refuse(ctx, "the device-side wide collapse"); }
