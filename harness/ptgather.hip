// ptgather — how fast can gfx950 serve the ACCESS PATTERN of the traversal kernel?  A measured ceiling to hold
// k_trace_fast_persistent against: the kernel moves too few HBM bytes for the HBM roofline to
// bind it, and its VALU issue rate is at 0.69 of peak; what is left is the vector-memory path serving, per
// lane, a DEPENDENT chain of 16-byte loads from a cache line of the lane's own (a node), the next line known
// only once the data has arrived.  This program runs exactly that and nothing else: every active lane walks a
// random cycle through a table of 128-byte records (7, 4 or 1 dwordx4 loads per record, the next index taken
// from the data), at the traversal kernel's occupancy (6 blocks of 256 threads per CU), with all 64 or only some
// lanes of a wave active.  Prints one JSON line per configuration: records per second.
//   ptgather [--records N] [--steps N]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

template <int LOADS>
__global__ __launch_bounds__(256) void k_walk(const float4* __restrict__ table, uint32_t n_records, uint32_t steps,
                                              uint32_t active_lanes, float* __restrict__ out) {
    const uint32_t gtid = blockIdx.x * 256u + threadIdx.x;
    if ((threadIdx.x & 63u) >= active_lanes) return;
    uint32_t idx = (gtid * 2654435761u) % n_records;
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; ++s) {
        const float4* p = table + (size_t)idx * 8u;
        float4 v[LOADS];
#pragma unroll
        for (int k = 0; k < LOADS; ++k) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < LOADS; ++k) acc += v[k].y;
        idx = __float_as_uint(v[LOADS - 1].x);          // the last load carries the link: every load is waited for
    }
    out[gtid] = acc;
}

int main(int argc, char** argv) {
    uint32_t n_records = 1u << 20, steps = 256;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--records") n_records = (uint32_t)std::atol(argv[i + 1]);
        else if (k == "--steps") steps = (uint32_t)std::atol(argv[i + 1]);
        else { std::fprintf(stderr, "usage: ptgather [--records N] [--steps N]\n"); return 2; }
    }
    if (n_records < 2 || n_records > (1u << 24) || steps < 1 || steps > (1u << 16)) { std::fprintf(stderr, "ptgather: arguments out of range\n"); return 2; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 6u, threads = blocks * 256u;
    // one random cycle over all records (Sattolo), the link stored in .x of every float4 of a record
    std::vector<uint32_t> next(n_records);
    for (uint32_t i = 0; i < n_records; ++i) next[i] = i;
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    for (uint32_t i = n_records - 1; i > 0; --i) std::swap(next[i], next[(uint32_t)(rnd() % i)]);
    std::vector<float4> table((size_t)n_records * 8u);
    for (uint32_t i = 0; i < n_records; ++i)
        for (int k = 0; k < 8; ++k) {
            float link;
            std::memcpy(&link, &next[i], 4);
            table[(size_t)i * 8u + k] = make_float4(link, 1.0f, 0.0f, 0.0f);
        }
    float4* d_table = nullptr;
    float* d_out = nullptr;
    CHECK(hipMalloc(&d_table, table.size() * sizeof(float4)));
    CHECK(hipMalloc(&d_out, (size_t)threads * sizeof(float)));
    CHECK(hipMemcpy(d_table, table.data(), table.size() * sizeof(float4), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int loads[3] = {7, 4, 1};
    const uint32_t lanes[3] = {64, 28, 16};
    for (int li = 0; li < 3; ++li)
        for (int ai = 0; ai < 3; ++ai) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {        // first repetition warms the caches
                CHECK(hipEventRecord(e0, 0));
                if (loads[li] == 7) hipLaunchKernelGGL(k_walk<7>, dim3(blocks), dim3(256), 0, 0, d_table, n_records, steps, lanes[ai], d_out);
                else if (loads[li] == 4) hipLaunchKernelGGL(k_walk<4>, dim3(blocks), dim3(256), 0, 0, d_table, n_records, steps, lanes[ai], d_out);
                else hipLaunchKernelGGL(k_walk<1>, dim3(blocks), dim3(256), 0, 0, d_table, n_records, steps, lanes[ai], d_out);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.0f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double walkers = (double)threads / 64.0 * lanes[ai];
            const double recs = walkers * steps;
            std::printf("{\"loads_per_record\": %d, \"active_lanes\": %u, \"records\": %u, \"table_MB\": %.0f, \"steps\": %u, \"ms\": %.3f, "
                        "\"Grecords_per_s\": %.2f, \"lane_loads_per_ns\": %.2f, \"ns_per_dependent_step\": %.0f}\n",
                        loads[li], lanes[ai], n_records, table.size() * 16.0 / 1e6, steps, best, recs / best / 1e6,
                        recs * loads[li] / best / 1e6, best * 1e6 / steps);
        }
    CHECK(hipFree(d_table));
    CHECK(hipFree(d_out));
    return 0;
}
