// host_gen.cpp — synthetic workloads of SURVEY.md §8(d) (harness helpers, no reference
// counterpart: the reference ships no scene and takes rays only from its own camera).
#include <cmath>
#include <cstring>

#include "../../include/ptcore.h"
#include "guard.h"
#include "../../include/pt_rng.h"

static int pt_gen_soup_body(uint64_t seed, uint32_t n, const float lo[3], const float hi[3],
                           float edge_scale, float log_scale_hi, uint32_t n_material_cycle,
                           pt_triangle* out) {
    if (!lo || !hi || (!out && n) || !(edge_scale > 0.0f)) return PT_ERR_INVALID;
    const bool log_uniform = log_scale_hi > edge_scale;
    const float ratio = log_uniform ? log_scale_hi / edge_scale : 1.0f;
    for (uint32_t i = 0; i < n; ++i) {
        pt_rng r;
        pt_rng_init(&r, seed, i, 0u, 0u);          // one stream per triangle
        pt_triangle& t = out[i];
        std::memset(&t, 0, sizeof(t));
        float c[3], e1[3], e2[3];
        for (int k = 0; k < 3; ++k) c[k] = lo[k] + pt_rng_f32(&r) * (hi[k] - lo[k]);
        float s = edge_scale;
        if (log_uniform) s = edge_scale * pt_expf(pt_rng_f32(&r) * pt_logf(ratio));
        for (int k = 0; k < 3; ++k) e1[k] = (pt_rng_f32(&r) * 2.0f - 1.0f) * s;
        for (int k = 0; k < 3; ++k) e2[k] = (pt_rng_f32(&r) * 2.0f - 1.0f) * s;
        for (int k = 0; k < 3; ++k) { t.a[k] = c[k]; t.b[k] = c[k] + e1[k]; t.c[k] = c[k] + e2[k]; }
        t.material_index = n_material_cycle ? 1u + (i % n_material_cycle) : 0u;
        t.flags = 0u;
    }
    return PT_OK;
}

extern "C" int pt_gen_soup(uint64_t seed, uint32_t n, const float lo[3], const float hi[3],
                           float edge_scale, float log_scale_hi, uint32_t n_material_cycle,
                           pt_triangle* out) {
    return ptc_noexcept_code([&] { return pt_gen_soup_body(seed, n, lo, hi, edge_scale, log_scale_hi, n_material_cycle, out); });
}

// Camera::new + generate_ray at pixel centres (camera.rs:11-28), row-major x + y*W.
static int pt_gen_primary_rays_body(const pt_camera* cam, pt_ray* out) {
    if (!cam || !out || cam->width == 0 || cam->height == 0) return PT_ERR_INVALID;
    const float w = (float)cam->width, h = (float)cam->height;
    const float sx = 1.0f / (w + 1.0f), sy = 1.0f / (h + 1.0f);
    const float sensor0 = cam->sensor_width, sensor1 = cam->sensor_width * h / w;
    for (uint32_t y = 0; y < cam->height; ++y)
        for (uint32_t x = 0; x < cam->width; ++x) {
            pt_ray& r = out[(size_t)x + (size_t)y * cam->width];
            r.o[0] = r.o[1] = r.o[2] = 0.0f; r.pad0 = 0.0f; r.pad1 = 0.0f;
            r.d[0] = (((float)x + 0.5f) * sx - 0.5f) * sensor0;
            r.d[1] = -(((float)y + 0.5f) * sy - 0.5f) * sensor1;
            r.d[2] = -cam->focal_length;
        }
    return PT_OK;
}

extern "C" int pt_gen_primary_rays(const pt_camera* cam, pt_ray* out) {
    return ptc_noexcept_code([&] { return pt_gen_primary_rays_body(cam, out); });
}
