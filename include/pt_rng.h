/* pt_rng.h — counter-based per-pixel random stream shared by host and device.
 *
 * Replaces the reference's thread-local, entropy-seeded xoshiro256++ (rng.rs:7-11), which makes
 * the reference non-reproducible and ties sample values to rayon's work stealing.  Here every
 * (seed, pixel, sample) owns an independent stream and draw number `i` of that stream is a pure
 * function — Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter =
 * (i >> 2, pixel, sample, PT_RNG_DOMAIN), word i & 3 — so a GPU lane and a CPU thread that render
 * the same sample draw identical numbers no matter how work is scheduled or sharded.
 *
 * The DISTRIBUTIONS drawn are the reference's (A.1/A.2 of SURVEY.md):
 *   Standard f32            rand 0.8.5   (u32 >> 8) * 2^-24                 material.rs:275, scene.rs:344
 *   Uniform<f32>(lo, hi)    rand 0.8.5   ([1,2) mantissa trick - 1) * scale + lo     main.rs:83-84
 *   UnitSphere              rand_distr 0.4.3  Marsaglia 1972 rejection       material.rs:168
 *   StandardNormal          rand_distr 0.4.3 uses a f64 ziggurat whose tables are not available
 *                           offline; replaced by an inverse-CDF sampler with the same
 *                           distribution (one u32 per variate)                main.rs:77-78
 * Draw ORDER per sample follows main.rs:77-84 and then, per path iteration, material.rs:275,
 * material.rs:168, scene.rs:344.
 */
#ifndef PT_RNG_H
#define PT_RNG_H

#include "pt_math.h"

#define PT_RNG_DOMAIN 0x70747263u /* "ptrc" */

typedef struct { uint32_t v[4]; } pt_u32x4;

PT_HD pt_u32x4 pt_philox4x32_10(pt_u32x4 ctr, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr.v[0], c1 = ctr.v[1], c2 = ctr.v[2], c3 = ctr.v[3];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    pt_u32x4 out;
    out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

/* One stream = one (seed, pixel, sample).  `index` is the number of u32 words drawn so far; it is
 * the only state that has to survive between kernels (4 bytes per path). */
typedef struct {
    uint32_t k0, k1;      /* seed */
    uint32_t pixel;
    uint32_t sample;
    uint32_t index;       /* next draw number */
    uint32_t cached;      /* block number held in `block`, 0xFFFFFFFF = none */
    pt_u32x4 block;
} pt_rng;

PT_HD void pt_rng_init(pt_rng* r, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t index) {
    r->k0 = (uint32_t)seed;
    r->k1 = (uint32_t)(seed >> 32);
    r->pixel = pixel;
    r->sample = sample;
    r->index = index;
    r->cached = 0xFFFFFFFFu;
    r->block.v[0] = r->block.v[1] = r->block.v[2] = r->block.v[3] = 0u;
}

PT_HD uint32_t pt_rng_u32(pt_rng* r) {
    uint32_t blk = r->index >> 2;
    if (blk != r->cached) {
        pt_u32x4 c;
        c.v[0] = blk; c.v[1] = r->pixel; c.v[2] = r->sample; c.v[3] = PT_RNG_DOMAIN;
        r->block = pt_philox4x32_10(c, r->k0, r->k1);
        r->cached = blk;
    }
    uint32_t w = r->index & 3u;
    r->index += 1u;
    /* Word w of the block by masks: no dynamic register index, and nothing the compiler can turn into a load from a
     * selected ADDRESS (it did that to a chain of `w == k ? v[k] : x`, which put the whole generator into scratch
     * memory: 44 bytes stored per lane and draw loop, each of them a write that reached HBM). */
    const uint32_t b0 = r->block.v[0], b1 = r->block.v[1], b2 = r->block.v[2], b3 = r->block.v[3];
    const uint32_t lo = b0 ^ ((b0 ^ b1) & (0u - (w & 1u)));       /* w odd: b1, else b0 */
    const uint32_t hi = b2 ^ ((b2 ^ b3) & (0u - (w & 1u)));       /* w odd: b3, else b2 */
    return lo ^ ((lo ^ hi) & (0u - (w >> 1)));
}

/* rand 0.8.5 `Standard` for f32: 24 random bits, [0, 1). */
PT_HD float pt_rng_f32(pt_rng* r) {
    return (float)(pt_rng_u32(r) >> 8) * 5.9604644775390625e-8f;   /* 2^-24 */
}

/* rand 0.8.5 `Uniform::new(lo, hi)` for f32, sample_single path via the stored `scale`:
 * 23 random mantissa bits in [1, 2), minus 1, times scale, plus lo (two roundings, no FMA). */
PT_HD float pt_uniform_scale(float lo, float hi) {
    float scale = hi - lo;
    const float max_rand = 1.0f - 1.1920928955078125e-7f;       /* 1 - f32::EPSILON */
    while (!(scale * max_rand + lo < hi)) {
        scale = pt_u2f(pt_f2u(scale) - 1u);                      /* next float toward zero */
    }
    return scale;
}
PT_HD float pt_rng_uniform(pt_rng* r, float lo, float scale) {
    float v12 = pt_u2f(0x3F800000u | (pt_rng_u32(r) >> 9));
    float v01 = v12 - 1.0f;
    return v01 * scale + lo;
}

/* rand_distr 0.4.3 `UnitSphere` (Marsaglia 1972).  Uniform(-1, 1) has scale exactly 2. */
PT_HD void pt_rng_unit_sphere(pt_rng* r, float out[3]) {
    for (;;) {
        float x1 = pt_rng_uniform(r, -1.0f, 2.0f);
        float x2 = pt_rng_uniform(r, -1.0f, 2.0f);
        float sum = x1 * x1 + x2 * x2;
        if (sum >= 1.0f) continue;
        float factor = 2.0f * pt_sqrt(1.0f - sum);
        out[0] = x1 * factor;
        out[1] = x2 * factor;
        out[2] = 1.0f - 2.0f * sum;
        return;
    }
}

/* Standard normal by inverse CDF on a 23-bit mid-point grid: (k + 0.5) * 2^-23 is exact in
 * binary32 for k < 2^23 and never 0 or 1. */
PT_HD float pt_rng_normal(pt_rng* r) {
    float p = ((float)(pt_rng_u32(r) >> 9) + 0.5f) * 1.1920928955078125e-7f;
    return pt_norm_ppf(p);
}

#endif /* PT_RNG_H */
