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

// Round 3: the COOPERATIVE pattern.  The same walkers, but a record is fetched by the 8 lanes of its owner's group in ONE
// instruction (lane l of a group reads the 16-byte piece l & 7 of the record the group's j-th lane wants, j = 0 .. 7: eight
// instructions fetch the group's eight records, each instruction touching one 128-byte line per group instead of eight).
// If the vector L1 charges per distinct line of an instruction rather than per lane, this serves records several times
// faster than the direct pattern; the pieces would then have to travel to their owner through LDS.  The link comes back to
// the owner by a cross-lane read (ds_bpermute: the LDS crossbar, no L1 traffic).
__global__ __launch_bounds__(256) void k_walk_coop(const float4* __restrict__ table, uint32_t n_records, uint32_t steps, float* __restrict__ out) {
    const uint32_t gtid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, piece = lane & 7u, leader = lane & ~7u;
    uint32_t idx = (gtid * 2654435761u) % n_records;
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; ++s) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t want = (uint32_t)__shfl((int)idx, (int)(leader + j), 64);
            v[j] = table[(size_t)want * 8u + piece];
        }
        uint32_t next = idx;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc += v[j].y;
            const uint32_t link = (uint32_t)__shfl((int)__float_as_uint(v[j].x), (int)leader, 64);     // piece 0 of record j sits in the leader lane
            if (piece == (uint32_t)j) next = link;
        }
        idx = next;
    }
    out[gtid] = acc;
}

// The cooperative pattern WITH the hand-over to the owner: the pieces go through a per-wave staging area in LDS and the
// owner reads its seven back (ds_read_b128), which is what a traversal kernel would have to do.  MODE 0: the eight loads
// land in VGPRs and are written to LDS in two batches of four (4 KB of staging per wave).  MODE 1: global_load_lds_dwordx4
// writes all eight straight into LDS (lane-linear: exactly the [group][piece] layout; 8 KB per wave, no VGPRs).
// MODE 2: four by global_load_lds, four through VGPRs, 4 KB per wave, one memory latency per step.
template <int MODE>
__global__ __launch_bounds__(256) void k_walk_coop_lds(const float4* __restrict__ table, uint32_t n_records, uint32_t steps, float* __restrict__ out) {
    constexpr int SLOTS = MODE == 1 ? 8 : 4;
    __shared__ float4 stage_all[4][SLOTS * 65];      // rows of 64 pieces, one float4 of padding: an owner's reads fall in 16 different banks per 16 lanes
    float4* stage = stage_all[threadIdx.x >> 6];
    const uint32_t gtid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, piece = lane & 7u, leader = lane & ~7u;
    uint32_t idx = (gtid * 2654435761u) % n_records;
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; ++s) {
        float4 r[7];
        uint32_t want[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) want[j] = (uint32_t)__shfl((int)idx, (int)(leader + j), 64);
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds(table + (size_t)want[j] * 8u + piece, (__attribute__((address_space(3))) void*)(stage + j * 65), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < 7; ++p) r[p] = stage[piece * 65u + leader + p];
        } else {
            float4 v[4];
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_global_load_lds(table + (size_t)want[j] * 8u + piece, (__attribute__((address_space(3))) void*)(stage + j * 65), 16, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = table[(size_t)want[4 + j] * 8u + piece];
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = table[(size_t)want[j] * 8u + piece];
#pragma unroll
                for (int j = 0; j < 4; ++j) stage[j * 65 + lane] = v[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = table[(size_t)want[4 + j] * 8u + piece];
            }
            __builtin_amdgcn_wave_barrier();
            if (piece < 4u) {
#pragma unroll
                for (int p = 0; p < 7; ++p) r[p] = stage[piece * 65u + leader + p];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) stage[j * 65 + lane] = v[j];
            __builtin_amdgcn_wave_barrier();
            if (piece >= 4u) {
#pragma unroll
                for (int p = 0; p < 7; ++p) r[p] = stage[(piece - 4u) * 65u + leader + p];
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int p = 0; p < 7; ++p) acc += r[p].y;
        idx = __float_as_uint(r[6].x);
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
    {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_walk_coop, dim3(blocks), dim3(256), 0, 0, d_table, n_records, steps, d_out);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double recs = (double)threads * steps;
        std::printf("{\"pattern\": \"cooperative: 8 lanes x 16 B per record, 8 records per instruction\", \"loads_per_record\": 8, \"active_lanes\": 64, \"records\": %u, "
                    "\"table_MB\": %.0f, \"steps\": %u, \"ms\": %.3f, \"Grecords_per_s\": %.2f, \"lane_loads_per_ns\": %.2f, \"ns_per_dependent_step\": %.0f}\n",
                    n_records, table.size() * 16.0 / 1e6, steps, best, recs / best / 1e6, recs * 8.0 / best / 1e6, best * 1e6 / steps);
    }
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k_walk_coop_lds<0>, dim3(blocks), dim3(256), 0, 0, d_table, n_records, steps, d_out);
            else if (mode == 1) hipLaunchKernelGGL(k_walk_coop_lds<1>, dim3(blocks), dim3(256), 0, 0, d_table, n_records, steps, d_out);
            else hipLaunchKernelGGL(k_walk_coop_lds<2>, dim3(blocks), dim3(256), 0, 0, d_table, n_records, steps, d_out);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        // the walk must have followed the cycle: out[] holds steps * 7 for every thread (each piece's .y is 1)
        std::vector<float> h(threads);
        CHECK(hipMemcpy(h.data(), d_out, (size_t)threads * sizeof(float), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (float x : h) bad += x != (float)(steps * 7u);
        const double recs = (double)threads * steps;
        const char* names[3] = {"cooperative + LDS hand-over, loads via VGPRs, 4 KB staging per wave", "cooperative + LDS hand-over, global_load_lds, 8 KB staging per wave",
                                "cooperative + LDS hand-over, 4 global_load_lds + 4 via VGPRs, 4 KB staging per wave"};
        std::printf("{\"pattern\": \"%s\", \"active_lanes\": 64, \"records\": %u, \"table_MB\": %.0f, \"steps\": %u, \"ms\": %.3f, \"Grecords_per_s\": %.2f, "
                    "\"ns_per_dependent_step\": %.0f, \"threads_with_a_wrong_sum\": %zu}\n",
                    names[mode], n_records, table.size() * 16.0 / 1e6, steps, best, recs / best / 1e6, best * 1e6 / steps, bad);
    }
    CHECK(hipFree(d_table));
    CHECK(hipFree(d_out));
    return 0;
}
