// render.hip — placeholder until the wavefront integrator lands (next commit).
#include "device_types.h"

extern "C" int pt_render_dev(pt_ctx* ctx, const pt_scene*, const pt_camera*, uint32_t, uint32_t, uint32_t,
                             uint64_t, uint32_t, float*) {
    return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_render_dev: integrator not built yet");
}
extern "C" int pt_render(pt_ctx* ctx, const pt_scene*, const pt_camera*, uint32_t, uint32_t, uint32_t,
                         uint64_t, uint32_t, float*) {
    return ptc_fail(ctx, PT_ERR_UNSUPPORTED, "pt_render: integrator not built yet");
}
