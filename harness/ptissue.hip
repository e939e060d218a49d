// ptissue.hip — measured VALU issue ceiling of the chip, the third on-chip roofline bench.py quotes beside the vector-L1
// access ceiling of ptgather: how many wave64 vector instructions per second all SIMDs issue when every wave runs
// nothing but independent VALU work, at 1 .. 8 waves per SIMD, for the instruction kinds the traversal kernels are made of
// (v_fma_f32, v_min_f32 / v_max_f32, v_cndmask_b32, v_pk_mul_f32 / v_pk_add_f32).  Prints one JSON line per case.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(float* out, int iters, float a, float b) {
    float acc[8];
    f2 pacc[8];
    for (int i = 0; i < 8; ++i) { acc[i] = a * (float)(threadIdx.x + i); pacc[i] = f2{acc[i], acc[i] + b}; }
    const f2 pa = {a, b};
    unsigned long long mask = __ballot(a > (float)threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pacc[i]) : "v"(pa));
                if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "s"(mask));
                if (KIND == 6) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[(i + 4) & 7]) : "v"(a), "v"(b)); }
                if (KIND == 7) { asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc"); }
                if (KIND == 8) { asm volatile("v_cmp_lt_f32_e64 %3, %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(acc[i]) : "v"(a), "v"(b), "s"(mask)); }
                if (KIND == 9) asm volatile("v_min_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                // round 3: the scalar forms of what the triangle test and the packed slab arithmetic are made of
                if (KIND == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 13) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
                if (KIND == 14) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
                if (KIND == 15) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(acc[i]), "v"(a) : "vcc");
                if (KIND == 16) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 17) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 18) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i]));
                if (KIND == 19) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pacc[i]) : "v"(pa));
                if (KIND == 20) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pacc[i]) : "v"(pa));
                if (KIND == 21) asm volatile("v_min_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 22) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 23) asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 24) asm volatile("v_cmp_ge_f32_e64 %2, %0, %1" : : "v"(acc[i]), "v"(a), "s"(mask));
                if (KIND == 25) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a));      // two distinct VGPR sources
                if (KIND == 26) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "s"(a));            // one source in an SGPR
                if (KIND == 27) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 28) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 29) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(acc[i]) : "v"(a) : "vcc");
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += acc[i] + pacc[i].x + pacc[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;          // never true: keeps the chains alive
}

template <int KIND>
void run(const char* name, int cus, float* d_out) {
    const int iters = 4096;
    for (int wps = 2; wps <= 8; wps *= 4) {
        const int blocks = cus * wps;                    // one 256-thread block = one wave on each of a CU's four SIMDs
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 64, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr = (double)blocks * 4.0 * iters * 64.0;
        std::printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s\": %.4g, \"per_simd_cycles_per_instr_at_2.4GHz\": %.3f}\n",
                    name, wps, ms, wave_instr / (ms * 1e-3), (ms * 1e-3 * 2.4e9) / ((double)wps * iters * 64.0));
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    float* d_out = nullptr;
    CHECK(hipMalloc((void**)&d_out, 4096));
    run<0>("v_fma_f32", prop.multiProcessorCount, d_out);
    run<1>("v_max_f32", prop.multiProcessorCount, d_out);
    run<2>("v_cndmask_b32", prop.multiProcessorCount, d_out);
    run<3>("v_pk_mul_f32", prop.multiProcessorCount, d_out);
    run<4>("v_max3_f32", prop.multiProcessorCount, d_out);
    run<5>("v_cndmask_b32_e64 (sgpr pair)", prop.multiProcessorCount, d_out);
    run<6>("v_cndmask_b32 vcc + v_fma_f32 (pair counted as 1)", prop.multiProcessorCount, d_out);
    run<7>("v_cmp_lt_f32 vcc + v_cndmask_b32 vcc (pair counted as 1)", prop.multiProcessorCount, d_out);
    run<8>("v_cmp_lt_f32_e64 + v_cndmask_b32_e64 (pair counted as 1)", prop.multiProcessorCount, d_out);
    run<9>("v_min_u32", prop.multiProcessorCount, d_out);
    run<10>("v_mov_b32", prop.multiProcessorCount, d_out);
    run<11>("v_add_u32", prop.multiProcessorCount, d_out);
    run<12>("v_mul_f32", prop.multiProcessorCount, d_out);
    run<13>("v_add_f32", prop.multiProcessorCount, d_out);
    run<14>("v_sub_f32", prop.multiProcessorCount, d_out);
    run<15>("v_cmp_ge_f32 vcc", prop.multiProcessorCount, d_out);
    run<16>("v_xor_b32", prop.multiProcessorCount, d_out);
    run<17>("v_or3_b32", prop.multiProcessorCount, d_out);
    run<18>("v_rcp_f32", prop.multiProcessorCount, d_out);
    run<19>("v_pk_add_f32", prop.multiProcessorCount, d_out);
    run<20>("v_pk_fma_f32", prop.multiProcessorCount, d_out);
    run<21>("v_min_f32", prop.multiProcessorCount, d_out);
    run<22>("v_min3_f32", prop.multiProcessorCount, d_out);
    run<23>("v_and_b32", prop.multiProcessorCount, d_out);
    run<24>("v_cmp_ge_f32_e64 sgpr pair", prop.multiProcessorCount, d_out);
    run<25>("v_fma_f32 (two distinct VGPR sources)", prop.multiProcessorCount, d_out);
    run<26>("v_mul_f32 (SGPR source)", prop.multiProcessorCount, d_out);
    run<27>("v_lshl_add_u32", prop.multiProcessorCount, d_out);
    run<28>("v_div_fixup_f32", prop.multiProcessorCount, d_out);
    run<29>("v_div_scale_f32", prop.multiProcessorCount, d_out);
    CHECK(hipFree(d_out));
    return 0;
}
