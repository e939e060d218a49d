/* pt_math.h — deterministic binary32 math shared by host and device.
 *
 * The reference (Rust, no fast-math, no FMA contraction) calls the platform libm for
 * exp / acos / sin / powf(.,1.5) / ln (color.rs:78-81, scene.rs:291-293,
 * hosek_wilkie.rs:312-319,371-374).  Host libm and device OCML differ in the last ulp, so a
 * CPU checker and the HIP kernels could never agree bit-for-bit on an image.  Every function here
 * is therefore written with IEEE-754 add/sub/mul/div/sqrt and integer bit operations only, all of
 * which are correctly rounded on x86-64 and on gfx950 — the same source gives the same bits on
 * both, provided the translation unit is built with -ffp-contract=off (no FMA fusion) and without
 * fast-math.  Accuracy is <= ~2 ulp against the true functions (tests/test_math.py pins that
 * against float64 references), i.e. far below the image tolerance stated in DESIGN.md.
 *
 * Algorithms: classic Cody–Waite range reduction + minimax polynomials in the style of the
 * public-domain Cephes single-precision library; AS241 (Wichura 1988, PPND7) for the inverse
 * normal CDF.  Nothing here comes from the reference repository.
 */
#ifndef PT_MATH_H
#define PT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

PT_HD uint32_t pt_f2u(float f) {
    union { float f; uint32_t u; } c; c.f = f; return c.u;
}
PT_HD float pt_u2f(uint32_t u) {
    union { float f; uint32_t u; } c; c.u = u; return c.f;
}

PT_HD float pt_inf() { return pt_u2f(0x7F800000u); }
PT_HD float pt_nan() { return pt_u2f(0x7FC00000u); }

/* Rust `f32::is_sign_positive`: sign BIT clear (so -0.0 and negative NaNs are "negative"). */
PT_HD bool pt_sign_positive(float x) { return (pt_f2u(x) >> 31) == 0u; }

/* Rust `f32::min` / `f32::max`: IEEE minNum/maxNum — the non-NaN operand wins. */
#if defined(__HIP_DEVICE_COMPILE__)
PT_HD float pt_fmin(float a, float b) { return __builtin_fminf(a, b); }   /* v_min_f32, IEEE mode */
PT_HD float pt_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
#else
/* Same value as minNum/maxNum for every input (only the sign of a zero result may differ, and no
 * caller observes that: results are only compared, subtracted or multiplied by finite values). */
PT_HD float pt_fmin(float a, float b) { return (b != b || a < b) ? a : b; }
PT_HD float pt_fmax(float a, float b) { return (b != b || a > b) ? a : b; }
#endif

PT_HD float pt_sqrt(float x) { return __builtin_sqrtf(x); }
PT_HD float pt_floor(float x) { return __builtin_floorf(x); }
PT_HD float pt_trunc(float x) { return __builtin_truncf(x); }
PT_HD float pt_fabs(float x) { return __builtin_fabsf(x); }

/* Rust `x as usize` for f32: saturating, NaN -> 0.  Clamped to 2^31 so it fits uint32_t; every
 * call site compares the result with a small table length afterwards. */
PT_HD uint32_t pt_f2usize(float x) {
    if (!(x > 0.0f)) return 0u;              /* negatives, -0, NaN */
    if (x >= 2147483648.0f) return 0x80000000u;
    return (uint32_t)x;                       /* truncation toward zero */
}

/* Rust `x % 1.0` (fmodf with divisor 1): exact. */
PT_HD float pt_fract_rem(float x) { return x - pt_trunc(x); }

/* 2^n for integer n in [-126, 127]. */
PT_HD float pt_pow2i(int n) { return pt_u2f((uint32_t)(n + 127) << 23); }

/* e^x.  Cody–Waite reduction x = n ln2 + r, |r| <= ln2/2, degree-5 minimax for (e^r-1-r)/r^2. */
PT_HD float pt_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return pt_inf();
    if (x < -103.97208404541015625f) return 0.0f;
    float fn = pt_floor(x * 1.44269504088896341f + 0.5f);
    int n = (int)fn;
    float r = x - fn * 0.693359375f;
    r = r - fn * -2.12194440e-4f;
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = p * r + 1.3981999507E-3f;
    p = p * r + 8.3334519073E-3f;
    p = p * r + 4.1665795894E-2f;
    p = p * r + 1.6666665459E-1f;
    p = p * r + 5.0000001201E-1f;
    p = p * z + r;
    p = p + 1.0f;
    /* scale in two exact steps so that n in [-150, 128] never overflows an intermediate */
    int n1 = n / 2;
    int n2 = n - n1;
    return (p * pt_pow2i(n1)) * pt_pow2i(n2);
}

/* ln x.  x = m 2^e with m in [sqrt(1/2), sqrt(2)), degree-8 minimax on m-1. */
PT_HD float pt_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return pt_nan();
    if (x == 0.0f) return -pt_inf();
    if (x == pt_inf()) return x;
    int e = 0;
    uint32_t u = pt_f2u(x);
    if ((u >> 23) == 0u) {                    /* subnormal: renormalise exactly */
        x = x * 33554432.0f;                  /* 2^25 */
        u = pt_f2u(x);
        e = -25;
    }
    e += (int)(u >> 23) - 126;
    float m = pt_u2f((u & 0x007FFFFFu) | 0x3F000000u);   /* [0.5, 1) */
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = y * m - 1.1514610310E-1f;
    y = y * m + 1.1676998740E-1f;
    y = y * m - 1.2420140846E-1f;
    y = y * m + 1.4249322787E-1f;
    y = y * m - 1.6668057665E-1f;
    y = y * m + 2.0000714765E-1f;
    y = y * m - 2.4999993993E-1f;
    y = y * m + 3.3333331174E-1f;
    y = y * m * z;
    float fe = (float)e;
    y = y + fe * -2.12194440e-4f;
    y = y - 0.5f * z;
    z = m + y;
    z = z + fe * 0.693359375f;
    return z;
}

/* asin on [-1, 1] (helper of pt_acosf). */
PT_HD float pt_asinf(float x) {
    float a = pt_fabs(x);
    if (!(a <= 1.0f)) return pt_nan();
    bool big = a > 0.5f;
    float z, s;
    if (big) { z = 0.5f * (1.0f - a); s = pt_sqrt(z); }
    else     { s = a; z = s * s; }
    float p = 4.2163199048E-2f;
    p = p * z + 2.4181311049E-2f;
    p = p * z + 4.5470025998E-2f;
    p = p * z + 7.4953002686E-2f;
    p = p * z + 1.6666752422E-1f;
    p = p * z * s + s;
    if (big) { p = p + p; p = 1.5707963267948966192f - p; }
    return (pt_f2u(x) >> 31) ? -p : p;
}

/* acos on [-1, 1]; acos(1) == 0 exactly. */
PT_HD float pt_acosf(float x) {
    if (!(pt_fabs(x) <= 1.0f)) return pt_nan();
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * pt_asinf(pt_sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * pt_asinf(pt_sqrt(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - pt_asinf(x);
}

/* sin for |x| < 8192 (three-part pi/4 reduction). */
PT_HD float pt_sinf(float x) {
    if (x != x) return x;
    bool neg = (pt_f2u(x) >> 31) != 0u;
    float a = pt_fabs(x);
    if (!(a < 8192.0f)) return (a == pt_inf()) ? pt_nan() : 0.0f;
    uint32_t j = (uint32_t)(a * 1.27323954473516f);      /* 4/pi */
    if (j & 1u) j += 1u;
    float y = (float)j;
    j &= 7u;
    if (j > 3u) { neg = !neg; j -= 4u; }
    float r = ((a - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = r * r;
    float v;
    if (j == 1u || j == 2u) {
        v = 2.443315711809948E-005f;
        v = v * z - 1.388731625493765E-003f;
        v = v * z + 4.166664568298827E-002f;
        v = v * z * z;
        v = v - 0.5f * z;
        v = v + 1.0f;
    } else {
        v = -1.9515295891E-4f;
        v = v * z + 8.3321608736E-3f;
        v = v * z - 1.6666654611E-1f;
        v = v * z * r;
        v = v + r;
    }
    return neg ? -v : v;
}

/* Cube root: exponent split by 3, quartic seed on [0.5, 1), two Newton steps. */
PT_HD float pt_cbrtf(float x) {
    if (x != x || x == 0.0f || pt_fabs(x) == pt_inf()) return x;
    bool neg = (pt_f2u(x) >> 31) != 0u;
    const float a = pt_fabs(x);
    int e = 0;
    uint32_t u = pt_f2u(a);
    if ((u >> 23) == 0u) { u = pt_f2u(a * 33554432.0f); e = -25; }
    e += (int)(u >> 23) - 126;
    float m = pt_u2f((u & 0x007FFFFFu) | 0x3F000000u);   /* [0.5, 1) */
    float y = -0.13466110473359520655053f;
    y = y * m + 0.54664601366395524503440f;
    y = y * m - 0.95438224771509446525043f;
    y = y * m + 1.1399983354717293273738f;
    y = y * m + 0.40238979564544752126924f;
    /* e = 3*q + rem with rem in {0,1,2} (floor division) */
    int q = (e >= 0) ? e / 3 : -((-e + 2) / 3);
    int rem = e - 3 * q;
    if (rem == 1) y = y * 1.25992104989487316477f;
    else if (rem == 2) y = y * 1.58740105196819947475f;
    y = y * pt_pow2i(q);
    y = y - (y - (a / (y * y))) * 0.33333333333333333333f;
    y = y - (y - (a / (y * y))) * 0.33333333333333333333f;
    return neg ? -y : y;
}

/* x^1.5 as x*sqrt(x): NaN for x < 0 (as powf), 0 -> 0, inf -> inf. */
PT_HD float pt_pow1p5(float x) { return x * pt_sqrt(x); }

/* Inverse standard-normal CDF, AS241 PPND7 (about 7 significant digits), p in (0, 1). */
PT_HD float pt_norm_ppf(float p) {
    float q = p - 0.5f;
    if (pt_fabs(q) <= 0.425f) {
        float r = 0.180625f - q * q;
        float num = ((59.109374720f * r + 159.29113202f) * r + 50.434271938f) * r + 3.3871327179f;
        float den = ((67.187563600f * r + 78.757757664f) * r + 17.895169469f) * r + 1.0f;
        return q * num / den;
    }
    float r = (q < 0.0f) ? p : 1.0f - p;
    r = pt_sqrt(-pt_logf(r));
    float v;
    if (r <= 5.0f) {
        r = r - 1.6f;
        float num = ((0.17023821103f * r + 1.3067284816f) * r + 2.7568153900f) * r + 1.4234372777f;
        float den = (0.12021132975f * r + 0.73700164250f) * r + 1.0f;
        v = num / den;
    } else {
        r = r - 5.0f;
        float num = ((0.017337203997f * r + 0.42868294337f) * r + 3.0812263860f) * r + 6.6579051150f;
        float den = (0.012258202635f * r + 0.24197894225f) * r + 1.0f;
        v = num / den;
    }
    return (q < 0.0f) ? -v : v;
}

#endif /* PT_MATH_H */
