// tonemap.hip — the step AFTER the hot path: what the reference's display does
// with the XYZ framebuffer, so that a render can be looked at without the winit/wgpu window.
//
//   renderer.rs:297-309   log-average luminance over the framebuffer, zones -> middle grey / white point
//   shader.wgsl:32-44     exposure scale, extended Reinhard on Y, XYZ -> linear sRGB (matrix rows as
//                         written there: the mat3x3 constructor takes COLUMNS and the shader multiplies
//                         a row vector, so component j is dot(xyz, column j))
//   renderer.rs:191,204   the surface is Bgra8Unorm: the linear value is clamped to [0,1] and quantised,
//                         with NO sRGB encoding (a quirk kept here)
//
// The reference adds ln(0.001 + |Y|) over all pixels sequentially in f32; here the sum is a two-level
// f64 reduction (block partials, then one block), so avg_luminance agrees to ~1e-6 relative, not
// bit for bit; the tolerance is written in tests/test_gpu_tonemap.py.
#include <hip/hip_runtime.h>

#include <cmath>

#include "device_types.h"
#include "../../include/pt_math.h"

#define PTC_BLOCK 256

namespace {

__global__ __launch_bounds__(PTC_BLOCK) void k_logsum(const float4* __restrict__ xyzw, uint32_t npix, uint32_t flags,
                                                     double* __restrict__ partial /* [gridDim.x][2] */) {
    __shared__ double s_sum[PTC_BLOCK], s_cnt[PTC_BLOCK];
    double sum = 0.0, cnt = 0.0;
    for (uint32_t i = blockIdx.x * PTC_BLOCK + threadIdx.x; i < npix; i += gridDim.x * PTC_BLOCK) {
        const float y = xyzw[i].y;
        const bool finite = pt_fabs(y) < pt_inf();          // false for NaN and +-inf
        if ((flags & PT_TONEMAP_SKIP_NONFINITE) && !finite) continue;
        sum += (double)pt_logf(0.001f + pt_fabs(y));        // renderer.rs:305
        cnt += 1.0;
    }
    s_sum[threadIdx.x] = sum; s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (uint32_t w = PTC_BLOCK / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) { s_sum[threadIdx.x] += s_sum[threadIdx.x + w]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s_sum[0]; partial[2 * blockIdx.x + 1] = s_cnt[0]; }
}

// shader.wgsl:33-43
__global__ __launch_bounds__(PTC_BLOCK) void k_tonemap(const float4* __restrict__ xyzw, uint32_t npix, float scale,
                                                      float max_lum_sq, uint32_t flags, float4* __restrict__ rgba,
                                                      uchar4* __restrict__ bgra8) {
    const uint32_t i = blockIdx.x * PTC_BLOCK + threadIdx.x;
    if (i >= npix) return;
    const float4 t = xyzw[i];
    float x = t.x, y = t.y, z = t.z;
    if ((flags & PT_TONEMAP_SKIP_NONFINITE) && !(pt_fabs(x) < pt_inf() && pt_fabs(y) < pt_inf() && pt_fabs(z) < pt_inf())) {
        x = 0.0f; y = 0.0f; z = 0.0f;
    }
    x *= scale; y *= scale; z *= scale;                       // xyz *= middle_gray / avg_luminance
    const float k = (1.0f + y / max_lum_sq) / (1.0f + y);     // extended Reinhard on Y
    x *= k; y *= k; z *= k;
    const float r = (x * 3.24096994f + y * -1.53738318f) + z * -0.49861076f;
    const float g = (x * -0.96924364f + y * 1.8759675f) + z * 0.04155506f;
    const float b = (x * 0.05563008f + y * -0.20397696f) + z * 1.05697151f;
    if (rgba) rgba[i] = make_float4(r, g, b, t.w);
    if (bgra8) {
        auto q = [](float v) -> unsigned char {               // Bgra8Unorm store: clamp, scale, round to nearest
            v = (v != v) ? 0.0f : pt_fmin(pt_fmax(v, 0.0f), 1.0f);
            return (unsigned char)(v * 255.0f + 0.5f);
        };
        bgra8[i] = make_uchar4(q(b), q(g), q(r), q(t.w));
    }
}

}  // namespace

static int pt_tonemap_dev_body(pt_ctx* ctx, const float* d_xyzw, uint32_t width, uint32_t height, float avg_zone,
                              float white_zone, uint32_t flags, float* d_rgba, uint8_t* d_bgra8, float stats_out[3]) {
    if (!ctx) return PT_ERR_INVALID;
    if (!d_xyzw || (!d_rgba && !d_bgra8 && !stats_out)) return ptc_fail(ctx, PT_ERR_INVALID, "pt_tonemap_dev: NULL argument");
    const uint64_t npix64 = (uint64_t)width * height;
    if (npix64 == 0 || npix64 > (1ull << 28)) return ptc_fail(ctx, PT_ERR_INVALID, "pt_tonemap_dev: image must have 1 .. 2^28 pixels");
    if (flags & ~(uint32_t)PT_TONEMAP_SKIP_NONFINITE) return ptc_fail(ctx, PT_ERR_INVALID, "pt_tonemap_dev: unknown flag");
    const uint32_t npix = (uint32_t)npix64;
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t rblocks = std::min<uint32_t>((npix + PTC_BLOCK - 1) / PTC_BLOCK, 1024u);
    int rc = ptc_reserve(ctx, &ctx->spill, &ctx->spill_bytes, (size_t)rblocks * 2 * sizeof(double), "tonemap partials");
    if (rc != PT_OK) return rc;
    double* d_partial = (double*)ctx->spill;
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_logsum, dim3(rblocks), dim3(PTC_BLOCK), 0, s, (const float4*)d_xyzw, npix, flags, d_partial);
    std::vector<double> partial((size_t)rblocks * 2);
    PTC_HIP(ctx, hipMemcpyAsync(partial.data(), d_partial, partial.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    PTC_HIP(ctx, hipStreamSynchronize(s));
    double sum = 0.0, cnt = 0.0;
    for (uint32_t b = 0; b < rblocks; ++b) { sum += partial[2 * b]; cnt += partial[2 * b + 1]; }
    // renderer.rs:307-309: avg = exp(4*sum / framebuffer.len()) with len = 4*pixels, i.e. exp(sum / pixels)
    const float avg_luminance = std::exp((float)(sum / (cnt > 0.0 ? cnt : 1.0)));
    const float max_luminance = avg_luminance * std::pow(2.0f, white_zone - avg_zone);
    const float middle_gray = 0.18f * std::pow(2.0f, avg_zone - 5.0f);
    if (stats_out) { stats_out[0] = avg_luminance; stats_out[1] = max_luminance; stats_out[2] = middle_gray; }
    if (d_rgba || d_bgra8) {
        hipLaunchKernelGGL(k_tonemap, dim3((npix + PTC_BLOCK - 1) / PTC_BLOCK), dim3(PTC_BLOCK), 0, s, (const float4*)d_xyzw, npix,
                           middle_gray / avg_luminance, max_luminance * max_luminance, flags, (float4*)d_rgba, (uchar4*)d_bgra8);
        PTC_HIP(ctx, hipGetLastError());
        PTC_HIP(ctx, hipStreamSynchronize(s));
    }
    return PT_OK;
}

extern "C" int pt_tonemap_dev(pt_ctx* ctx, const float* d_xyzw, uint32_t width, uint32_t height, float avg_zone,
                              float white_zone, uint32_t flags, float* d_rgba, uint8_t* d_bgra8, float stats_out[3]) {
    return ptc_guard(ctx, "pt_tonemap_dev", [&] { return pt_tonemap_dev_body(ctx, d_xyzw, width, height, avg_zone, white_zone, flags, d_rgba, d_bgra8, stats_out); });
}

static int pt_tonemap_body(pt_ctx* ctx, const float* xyzw, uint32_t width, uint32_t height, float avg_zone, float white_zone,
                          uint32_t flags, float* rgba, uint8_t* bgra8, float stats_out[3]) {
    if (!ctx) return PT_ERR_INVALID;
    if (!xyzw) return ptc_fail(ctx, PT_ERR_INVALID, "pt_tonemap: NULL argument");
    const size_t npix = (size_t)width * height;
    if (npix == 0 || npix > (1ull << 28)) return ptc_fail(ctx, PT_ERR_INVALID, "pt_tonemap: image must have 1 .. 2^28 pixels");
    PTC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t fb = npix * 16, bb = npix * 4;
    int rc = ptc_reserve(ctx, &ctx->io, &ctx->io_bytes, 2 * fb + bb, "tonemap staging");
    if (rc != PT_OK) return rc;
    char* base = (char*)ctx->io;
    PTC_HIP(ctx, hipMemcpy(base, xyzw, fb, hipMemcpyHostToDevice));
    rc = pt_tonemap_dev(ctx, (const float*)base, width, height, avg_zone, white_zone, flags, rgba ? (float*)(base + fb) : nullptr,
                        bgra8 ? (uint8_t*)(base + 2 * fb) : nullptr, stats_out);
    if (rc != PT_OK) return rc;
    if (rgba) PTC_HIP(ctx, hipMemcpy(rgba, base + fb, fb, hipMemcpyDeviceToHost));
    if (bgra8) PTC_HIP(ctx, hipMemcpy(bgra8, base + 2 * fb, bb, hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_tonemap(pt_ctx* ctx, const float* xyzw, uint32_t width, uint32_t height, float avg_zone, float white_zone,
                          uint32_t flags, float* rgba, uint8_t* bgra8, float stats_out[3]) {
    return ptc_guard(ctx, "pt_tonemap", [&] { return pt_tonemap_body(ctx, xyzw, width, height, avg_zone, white_zone, flags, rgba, bgra8, stats_out); });
}
