// shading.h — device-side spectral shading: what the reference evaluates per path vertex.
//
//   Color<SRGB>::reflectance_at_one     color.rs:41-54
//   Color<XYZ>::from_wavelength         color.rs:72-93
//   D65::sample_one                     material.rs:132-141
//   Texture::sample (Flat / Nearest / Bilinear)    material.rs:90-124
//   HosekWilkieSkyModel::radiance_one / solar_radiance_one and helpers   hosek_wilkie.rs:74-131,306-383
//   Scene::evaluate_sky                 scene.rs:284-305
//
// Every expression keeps the reference's association order (nalgebra dot = (x*x' + y*y') + z*z',
// lerp = a*(1-t) + b*t; a*b*c = (a*b)*c) and the file is built with -ffp-contract=off, so the CPU
// checker, which runs the same operations, produces the same bits.  Transcendentals come from
// include/pt_math.h on both sides.
#ifndef PTC_SHADING_H
#define PTC_SHADING_H

#include "device_types.h"
#include "../../include/pt_math.h"

#define PTC_NW 8                       /* COLOR_SAMPLES, main.rs:26 */

namespace ptc {

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

// wavelengths of one sample from its hero offset (main.rs:82-89)
__device__ __forceinline__ void wavelengths(float hero, float wl[PTC_NW]) {
    const float width = 830.0f - 360.0f;
#pragma unroll
    for (int i = 0; i < PTC_NW; ++i) wl[i] = 360.0f + (hero + width * ((float)i / (float)PTC_NW));
}

__device__ __forceinline__ float reflectance_at_one(const float* __restrict__ tab, float r, float g, float b,
                                                    float wl) {
    const float offset = wl - 360.0f;
    const float index = pt_floor(offset);
    const float alpha = offset - index;
    const uint32_t i = pt_f2usize(index);
    // TABLE.get(i).unwrap_or([0.0; 3]) for rows i and i+1.  The loads are UNCONDITIONAL (clamped index,
    // zero selected afterwards) so that the rows of all eight wavelengths can be in flight together
    // instead of one guarded round trip after another.
    const uint32_t i0 = i < 470u ? i : 470u, i1 = i < 469u ? i + 1u : 470u;
    float b0x = tab[3u * i0], b0y = tab[3u * i0 + 1u], b0z = tab[3u * i0 + 2u];
    float b1x = tab[3u * i1], b1y = tab[3u * i1 + 1u], b1z = tab[3u * i1 + 2u];
    if (!(i < 471u)) { b0x = 0.0f; b0y = 0.0f; b0z = 0.0f; }
    if (!(i < 470u)) { b1x = 0.0f; b1y = 0.0f; b1z = 0.0f; }
    const float w0 = 1.0f - alpha;
    const float bx = b0x * w0 + b1x * alpha;
    const float by = b0y * w0 + b1y * alpha;
    const float bz = b0z * w0 + b1z * alpha;
    return dot3(r, g, b, bx, by, bz);
}

__device__ __forceinline__ float d65_one(const float* __restrict__ tab, float wl) {
    const float offset = pt_floor(wl) - 300.0f;
    const float index = pt_floor(offset);
    const float alpha = offset - index;
    const uint32_t i = pt_f2usize(index);
    const uint32_t i0 = i < 530u ? i : 530u, i1 = i < 529u ? i + 1u : 530u;   // unconditional loads, as above
    float t0 = tab[i0], t1 = tab[i1];
    if (!(i < 531u)) t0 = 0.0f;
    if (!(i < 530u)) t1 = 0.0f;
    return t0 * (1.0f - alpha) + t1 * alpha;
}


// ---- LDS-resident copies of the two spectral tables (k_shade) ------------------------------------------
// Rows are padded to float4 and followed by one all-zero row, so that `TABLE.get(i).unwrap_or(0)`
// (color.rs:48-50, material.rs:139-140) is a clamped index and nothing else; a lane's eight wavelengths
// then cost 16 ds_read_b128 that never touch the vector-memory address path.
#define PTC_REFL_ROWS 471u
#define PTC_D65_ROWS 531u
struct SpectralLds {
    float4 refl[PTC_REFL_ROWS + 1u];
    float d65[PTC_D65_ROWS + 1u];
};
__device__ __forceinline__ void spectral_lds_fill(SpectralLds& t, const float* __restrict__ refl_tab,
                                                  const float* __restrict__ d65_tab, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i <= PTC_REFL_ROWS; i += nthreads)
        t.refl[i] = (i < PTC_REFL_ROWS) ? make_float4(refl_tab[3u * i], refl_tab[3u * i + 1u], refl_tab[3u * i + 2u], 0.0f)
                                        : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t i = tid; i <= PTC_D65_ROWS; i += nthreads) t.d65[i] = (i < PTC_D65_ROWS) ? d65_tab[i] : 0.0f;
}
__device__ __forceinline__ float reflectance_at_one(const SpectralLds& t, float r, float g, float b, float wl) {
    const float offset = wl - 360.0f;
    const float index = pt_floor(offset);
    const float alpha = offset - index;
    const uint32_t i = pt_f2usize(index);
    const float4 b0 = t.refl[i < PTC_REFL_ROWS ? i : PTC_REFL_ROWS];
    const float4 b1 = t.refl[i < PTC_REFL_ROWS - 1u ? i + 1u : PTC_REFL_ROWS];
    const float w0 = 1.0f - alpha;
    return dot3(r, g, b, b0.x * w0 + b1.x * alpha, b0.y * w0 + b1.y * alpha, b0.z * w0 + b1.z * alpha);
}
__device__ __forceinline__ float d65_one(const SpectralLds& t, float wl) {
    const float offset = pt_floor(wl) - 300.0f;
    const float index = pt_floor(offset);
    const float alpha = offset - index;
    const uint32_t i = pt_f2usize(index);
    const float t0 = t.d65[i < PTC_D65_ROWS ? i : PTC_D65_ROWS];
    const float t1 = t.d65[i < PTC_D65_ROWS - 1u ? i + 1u : PTC_D65_ROWS];
    return t0 * (1.0f - alpha) + t1 * alpha;
}

__device__ __forceinline__ float cmf_lobe(float wl, float weight, float mean, float s1, float s2) {
    const float t = (wl - mean) * (wl < mean ? s1 : s2);
    return weight * pt_expf(-0.5f * (t * t));
}
__device__ __forceinline__ void from_wavelength(float wl, float& X, float& Y, float& Z) {
    X = cmf_lobe(wl, 0.362f, 442.0f, 0.0624f, 0.0374f) + cmf_lobe(wl, 1.056f, 599.8f, 0.0264f, 0.0323f)
        + cmf_lobe(wl, -0.065f, 501.1f, 0.0490f, 0.0382f);
    Y = cmf_lobe(wl, 0.821f, 568.8f, 0.0213f, 0.0247f) + cmf_lobe(wl, 0.286f, 530.9f, 0.0613f, 0.0322f);
    Z = cmf_lobe(wl, 1.217f, 437.0f, 0.0845f, 0.0278f) + cmf_lobe(wl, 0.681f, 459.0f, 0.0385f, 0.0725f);
}

__device__ __forceinline__ float rem_euclid1(float x) {
    const float r = x - pt_trunc(x);           // x % 1.0, exact
    return (r < 0.0f) ? r + 1.0f : r;
}

// Texture::sample.  img == 0xFFFFFFFF: Flat(value)
__device__ __forceinline__ void texture_rgb(const DevSceneView& sc, uint32_t img, const float flat[3], float u,
                                            float v, float& r, float& g, float& b) {
    if (img == 0xFFFFFFFFu) { r = flat[0]; g = flat[1]; b = flat[2]; return; }
    const DevImage im = sc.images[img & ~PTC_IMG_BILINEAR];
    const float fx = rem_euclid1(u) * (float)im.width, fy = (1.0f - rem_euclid1(v)) * (float)im.height;
    const uint32_t x0 = pt_f2usize(pt_floor(fx)), y0 = pt_f2usize(pt_floor(fy));       // `as u32`: saturating, NaN -> 0
    const float* base = sc.texels + im.offset;
    auto texel = [&](uint32_t x, uint32_t y, float& tr, float& tg, float& tb) {
        const float* p = base + ((size_t)(x % im.width) + (size_t)(y % im.height) * im.width) * im.channels;
        if (im.channels == 3u) { tr = p[0]; tg = p[1]; tb = p[2]; }
        else { tr = tg = tb = p[0]; }
    };
    if (!(img & PTC_IMG_BILINEAR)) { texel(x0, y0, r, g, b); return; }                 // Nearest, material.rs:92-100
    // Bilinear, material.rs:101-123: p0 = p00 * (1 - tx) + p10 * tx, p1 = p01 * (1 - tx) + p11 * tx, p0 * (1 - ty) + p1 * ty
    const float tx = fx - pt_floor(fx), ty = fy - pt_floor(fy);
    float a[3], c[3], d[3], e[3];
    texel(x0, y0, a[0], a[1], a[2]);
    texel(x0 + 1u, y0, c[0], c[1], c[2]);
    texel(x0, y0 + 1u, d[0], d[1], d[2]);
    texel(x0 + 1u, y0 + 1u, e[0], e[1], e[2]);
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p0 = a[k] * (1.0f - tx) + c[k] * tx;
        const float p1 = d[k] * (1.0f - tx) + e[k] * tx;
        o[k] = p0 * (1.0f - ty) + p1 * ty;
    }
    r = o[0]; g = o[1]; b = o[2];
}

// ---- Hosek-Wilkie ---------------------------------------------------------------------------------
__device__ __forceinline__ float sq(float x) { return x * x; }
__device__ __forceinline__ float pow3(float x) { return x * (x * x); }
__device__ __forceinline__ float pow4(float x) { const float s = x * x; return s * s; }
__device__ __forceinline__ float pow5(float x) { const float s = x * x; return x * (s * s); }

__device__ __forceinline__ float hw_radiance_internal(const float* __restrict__ c, float cos_theta, float gamma,
                                                      float cos_gamma) {          // hosek_wilkie.rs:306-325
    const float exp_m = pt_expf(c[4] * gamma);
    const float ray_m = cos_gamma * cos_gamma;
    const float mie_m = (1.0f + cos_gamma * cos_gamma) / pt_pow1p5(1.0f + c[8] * c[8] - 2.0f * c[8] * cos_gamma);
    const float zenith = pt_sqrt(cos_theta);
    return (1.0f + c[0] * pt_expf(c[1] / (cos_theta + 0.01f)))
           * (c[2] + c[3] * exp_m + c[5] * ray_m + c[6] * mie_m + c[7] * zenith);
}

__device__ __forceinline__ float hw_radiance_one(const DevSky* __restrict__ sky, float cos_theta, float gamma,
                                                 float cos_gamma, float wl) {     // hosek_wilkie.rs:74-102
    const float band = (wl - 320.0f) / 40.0f;
    const uint32_t low = pt_f2usize(band);
    if (low >= 11u) return 0.0f;
    const float interp = pt_fract_rem(band);
    const float val_low = hw_radiance_internal(sky->configs[low], cos_theta, gamma, cos_gamma) * sky->radiances[low];
    if (interp < 1e-6f) return val_low;
    float result = (1.0f - interp) * val_low;
    if (low + 1u < 11u)
        result += interp * hw_radiance_internal(sky->configs[low + 1u], cos_theta, gamma, cos_gamma)
                  * sky->radiances[low + 1u];
    return result;
}

// sr_internal (hosek_wilkie.rs:327-345); `row` selects turb_low (0) or turb_low+1 (1)
__device__ __forceinline__ float hw_sr(const float* __restrict__ solar, uint32_t row, uint32_t wl, float elevation) {
    uint32_t pos = pt_f2usize(pt_cbrtf(2.0f * elevation / 3.14159265358979323846f) * 45.0f);
    if (pos > 44u) pos = 44u;
    const float break_x = pow3((float)pos / 45.0f) * 1.57079632679489661923f;
    const float x = elevation - break_x;
    const float* c = solar + ((wl * 2u + row) * 45u + pos) * 4u;
    float res = 0.0f, x_exp = 1.0f;
    res += x_exp * c[3]; x_exp *= x;
    res += x_exp * c[2]; x_exp *= x;
    res += x_exp * c[1]; x_exp *= x;
    res += x_exp * c[0];
    return res;
}

__device__ __forceinline__ float hw_solar_direct(const DevSceneView& sc, const DevSky* __restrict__ sky, float wl,
                                                 float elevation, float gamma) {  // hosek_wilkie.rs:347-383
    const float turb_frac = sky->turb_frac;
    uint32_t wl_low = pt_f2usize((wl - 320.0f) / 40.0f);
    if (wl_low > 9u) wl_low = 9u;
    const float wl_frac = (wl - 320.0f - (float)wl_low * 40.0f) / 40.0f;
    const float* S = sc.solar_tab;
    const float direct =
        (1.0f - turb_frac) * ((1.0f - wl_frac) * hw_sr(S, 0u, wl_low, elevation) + wl_frac * hw_sr(S, 0u, wl_low + 1u, elevation))
        + turb_frac * ((1.0f - wl_frac) * hw_sr(S, 1u, wl_low, elevation) + wl_frac * hw_sr(S, 1u, wl_low + 1u, elevation));
    float ld[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
        ld[i] = (1.0f - wl_frac) * sc.limb_tab[wl_low * 6u + i] + wl_frac * sc.limb_tab[(wl_low + 1u) * 6u + i];
    const float srs = pt_sinf(sky->solar_radius);
    const float ar2 = 1.0f / (srs * srs);
    const float sg = pt_sinf(gamma);
    const float scos = pt_sqrt(pt_fmax(1.0f - ar2 * sg * sg, 0.0f));
    const float dark = ld[0] + ld[1] * scos + ld[2] * sq(scos) + ld[3] * pow3(scos) + ld[4] * pow4(scos) + ld[5] * pow5(scos);
    return direct * dark;
}

// Scene::evaluate_sky (scene.rs:284-305); d is the UN-normalised ray direction
__device__ __forceinline__ void evaluate_sky(const DevSceneView& sc, const DevSky* __restrict__ sky, float dx,
                                             float dy, float dz, const float wl[PTC_NW], float out[PTC_NW]) {
    if (sky->kind == PT_SKY_D65) {
        const float power = sky->power;
        if (power == 0.0f) {
#pragma unroll
            for (int i = 0; i < PTC_NW; ++i) out[i] = 0.0f;
        } else {
#pragma unroll
            for (int i = 0; i < PTC_NW; ++i) out[i] = d65_one(sc.d65_tab, wl[i]) * power;
        }
        return;
    }
    const float cos_theta = dot3(dx, dy, dz, sky->zenith[0], sky->zenith[1], sky->zenith[2]);
    if (cos_theta > 0.0f) {
        const float theta = pt_acosf(pt_fmin(cos_theta, 1.0f));
        const float cos_gamma = dot3(dx, dy, dz, sky->sun[0], sky->sun[1], sky->sun[2]);
        const float gamma = pt_acosf(pt_fmax(pt_fmin(cos_gamma, 1.0f), -1.0f));
        if (gamma < sky->solar_radius) {
            for (int i = 0; i < PTC_NW; ++i) {                                    // hosek_wilkie.rs:114-131
                const float w = wl[i];
                float v = 0.0f;
                if (!(w < 320.0f || w > 720.0f)) {
                    const float direct = hw_solar_direct(sc, sky, w, 1.57079632679489661923f - theta, gamma);
                    const float inscattered = hw_radiance_one(sky, cos_theta, gamma, cos_gamma, w);
                    v = direct + inscattered;
                }
                out[i] = v;
            }
        } else {
            for (int i = 0; i < PTC_NW; ++i) out[i] = hw_radiance_one(sky, cos_theta, gamma, cos_gamma, wl[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < PTC_NW; ++i) out[i] = 0.0f;
    }
}

}  // namespace ptc
#endif
