// host_sky.cpp — one-off host initialisation the Rust host performs at scene load:
//   InitialisedSky::from_sky        scene.rs:105-129
//   HosekWilkieSkyModel::new        hosek_wilkie.rs:46-72
//   cook_configuration              hosek_wilkie.rs:144-229
//   cook_radiance_configuration     hosek_wilkie.rs:231-304
//   Material::default               material.rs:300-309
// A Rust caller passes its own cooked model through pt_sky; these entry points exist for hosts
// that are not the reference (the C++/Python harness in this repo).
//
// The reference writes the quintic Bernstein blend out eight times; here the six basis weights
//   (1-s)^5, 5(1-s)^4 s, 10(1-s)^3 s^2, 10(1-s)^2 s^3, 5(1-s) s^4, s^5
// are formed once with the reference's multiplication order (a*b*c*d is ((a*b)*c)*d) and reused,
// which leaves every product and sum — and so every output bit — unchanged.
#include <cmath>
#include <cstring>

#include "../../include/ptcore.h"
#include "guard.h"
#include "../../include/pt_math.h"
#include "tables.h"

namespace {

// llvm.powi semantics for small constant exponents: binary square-and-multiply.
inline float sq(float x) { return x * x; }
inline float pow3(float x) { return x * sq(x); }
inline float pow4(float x) { return sq(sq(x)); }
inline float pow5(float x) { return x * pow4(x); }

struct Bernstein5 {
    float w[6];
    explicit Bernstein5(float s) {
        const float r = 1.0f - s;
        w[0] = pow5(r);
        w[1] = 5.0f * pow4(r) * s;
        w[2] = 10.0f * pow3(r) * sq(s);
        w[3] = 10.0f * sq(r) * pow3(s);
        w[4] = 5.0f * r * pow4(s);
        w[5] = pow5(s);
    }
    // sum_k w[k]*e[k*stride], added left to right
    float blend(const float* e, int stride) const {
        float acc = w[0] * e[0];
        for (int k = 1; k < 6; ++k) acc = acc + w[k] * e[k * stride];
        return acc;
    }
};

inline void xform_dir(const float* m, const float v[3], float out[3]) {
    // nalgebra Matrix4::transform_vector: divide by (bottom row . v) when non-zero, then M3x3 * v
    float x = v[0], y = v[1], z = v[2];
    if (m) {
        float n = (m[3] * x + m[7] * y) + m[11] * z;
        if (n != 0.0f) { x = x / n; y = y / n; z = z / n; }
        out[0] = (m[0] * x + m[4] * y) + m[8] * z;
        out[1] = (m[1] * x + m[5] * y) + m[9] * z;
        out[2] = (m[2] * x + m[6] * y) + m[10] * z;
    } else {
        out[0] = x; out[1] = y; out[2] = z;
    }
    float len = std::sqrt((out[0] * out[0] + out[1] * out[1]) + out[2] * out[2]);
    out[0] = out[0] / len; out[1] = out[1] / len; out[2] = out[2] / len;   // .normalize()
}

}  // namespace

static int pt_sky_hosek_wilkie_body(float turbidity, float albedo, const float zenith[3],
                                   const float sun[3], const float* camera_transform, pt_sky* out) {
    if (!zenith || !sun || !out) return PT_ERR_INVALID;
    if (!(turbidity >= 1.0f && turbidity <= 10.0f)) return PT_ERR_INVALID;   // hosek_wilkie.rs:155,351-352
    std::memset(out, 0, sizeof(*out));
    out->kind = PT_SKY_HOSEK_WILKIE;
    xform_dir(camera_transform, zenith, out->zenith);                        // scene.rs:115-118
    xform_dir(camera_transform, sun, out->sun);
    const float cosz = (out->zenith[0] * out->sun[0] + out->zenith[1] * out->sun[1]) + out->zenith[2] * out->sun[2];
    const float elevation = std::asin(cosz);                                 // scene.rs:119
    if (!(elevation > 0.0f)) return PT_ERR_INVALID;                          // scene.rs:121

    out->turbidity = turbidity;
    out->solar_radius = (0.51f * (3.14159265358979323846f / 180.0f)) / 2.0f; // hosek_wilkie.rs:51

    const int ti = (int)pt_f2usize(turbidity);                               // `turbidity as usize`
    const float trem = turbidity - (float)ti;
    const Bernstein5 basis(std::cbrt(elevation / 1.57079632679489661923f));  // hosek_wilkie.rs:153,239

    // the four (albedo, turbidity) corners in the order the reference accumulates them
    const float corner_w[4] = {(1.0f - albedo) * (1.0f - trem), albedo * (1.0f - trem),
                               (1.0f - albedo) * trem, albedo * trem};
    const int corner_alb[4] = {0, 1, 0, 1};
    const int corner_turb[4] = {ti - 1, ti - 1, ti, ti};
    const int n_corners = (ti == 10) ? 2 : 4;                                // hosek_wilkie.rs:190,270

    for (int wl = 0; wl < 11; ++wl) {
        for (int i = 0; i < 9; ++i) {
            float acc = 0.0f;
            for (int c = 0; c < n_corners; ++c) {
                // DATASETS[wl][alb][turb][k][i], k stride 9
                const float* e = &ptc_tab_hw_datasets[(((wl * 2 + corner_alb[c]) * 10 + corner_turb[c]) * 6) * 9 + i];
                float term = corner_w[c] * basis.blend(e, 9);
                acc = (c == 0) ? term : acc + term;
            }
            out->configs[wl][i] = acc;
        }
        float acc = 0.0f;
        for (int c = 0; c < n_corners; ++c) {
            const float* e = &ptc_tab_hw_datasets_rad[((wl * 2 + corner_alb[c]) * 10 + corner_turb[c]) * 6];
            float term = corner_w[c] * basis.blend(e, 1);
            acc = (c == 0) ? term : acc + term;
        }
        out->radiances[wl] = acc;
    }
    return PT_OK;
}

extern "C" int pt_sky_hosek_wilkie(float turbidity, float albedo, const float zenith[3],
                                   const float sun[3], const float* camera_transform, pt_sky* out) {
    return ptc_noexcept_code([&] { return pt_sky_hosek_wilkie_body(turbidity, albedo, zenith, sun, camera_transform, out); });
}

extern "C" void pt_sky_d65(float power, pt_sky* out) {
    std::memset(out, 0, sizeof(*out));
    out->kind = PT_SKY_D65;
    out->power = power;
}

extern "C" void pt_material_default(pt_material* out) {
    std::memset(out, 0, sizeof(*out));
    out->diffuse.kind = PT_TEX_FLAT;
    out->diffuse.value[0] = out->diffuse.value[1] = out->diffuse.value[2] = 0.75f;
    out->emit_kind = PT_EMIT_NULL;
    out->absorption.kind = PT_TEX_FLAT;
    out->base_dissolve = 1.0f;
    out->dissolve.kind = PT_TEX_FLAT;
    out->dissolve.value[0] = 1.0f;
}
