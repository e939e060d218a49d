// ptrender_mgpu — the sample-split multi-GPU render of BASELINE configs[3] from a compiled host: one process per GPU, the
// scene replicated, rank g renders sample indices sample_range(spp, g, G) of every pixel into a PT_RENDER_SUM buffer in HBM,
// ONE ncclAllReduce(ncclFloat, ncclSum) over xGMI combines the W*H*4-float buffers, finalize_sums divides by the samples
// (main.rs:35-52,100-107 is the shared framebuffer this replaces).  The Rust host would do the same through
// the FFI crate of INTEGRATION.md; this is the C++ rank program the repository can build and test.
//
//   ptrender_mgpu scene.toml out_prefix [--gpus G] [--spp N] [--seed S]
//
// The launcher process never touches the GPU: it forks G rank processes BEFORE any HIP call (a forked child of a process
// that has initialised the GPU is not usable), hands them the path of a rendezvous file, and waits.  Rank 0 makes the
// ncclUniqueId and publishes it through that file (write + rename); every rank then joins with ncclCommInitRank.  Rank g
// uses GPU g.  Rank 0 writes out_prefix.pfm / .png and one JSON line with the timings (render seconds = slowest rank).
//
// Test/demo harness, not the product: the product is libptcore.so behind include/ptcore.h.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/ptcore.hpp"

namespace {

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, ncclGetErrorString(r_)); return 1; } } while (0)
int g_rank = 0;

void write_pfm(const std::string& path, const std::vector<float>& xyzw, uint32_t w, uint32_t h) {
    std::ofstream f(path, std::ios::binary);
    f << "PF\n" << w << " " << h << "\n-1.0\n";
    for (uint32_t y = h; y-- > 0;)
        for (uint32_t x = 0; x < w; ++x) f.write((const char*)&xyzw[4 * ((size_t)y * w + x)], 12);
}

int run_rank(int rank, int world, const std::string& scene_path, const std::string& out, long spp, uint64_t seed, const std::string& id_file) {
    g_rank = rank;
    using namespace pathtracer;
    using clock = std::chrono::steady_clock;
    const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // ---- rendezvous: rank 0 publishes the unique id --------------------------------------------------------------
    ncclUniqueId id;
    if (rank == 0) {
        NCCLCHECK(ncclGetUniqueId(&id));
        const std::string tmp = id_file + ".tmp";
        { std::ofstream f(tmp, std::ios::binary); f.write((const char*)&id, sizeof(id)); }
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) { std::perror("rename"); return 1; }
    } else {
        for (int tries = 0;; ++tries) {
            std::ifstream f(id_file, std::ios::binary);
            if (f && f.read((char*)&id, sizeof(id))) break;
            if (tries > 6000) { std::fprintf(stderr, "rank %d: no rendezvous file after 60 s\n", rank); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    int n_dev = 0;
    HIPCHECK(hipGetDeviceCount(&n_dev));
    if (rank >= n_dev) { std::fprintf(stderr, "rank %d: only %d GPUs visible (one rank per GPU)\n", rank, n_dev); return 1; }
    HIPCHECK(hipSetDevice(rank));
    ncclComm_t comm;
    NCCLCHECK(ncclCommInitRank(&comm, world, id, rank));
    try {
        Device dev(rank);
        const auto t0 = clock::now();
        Scene scene = Scene::load(dev, scene_path);
        const uint32_t w = scene.camera.resolution[0], h = scene.camera.resolution[1];
        const uint32_t samples = spp > 0 ? (uint32_t)spp : scene.camera.samples;
        const auto range = sample_range(samples, (uint32_t)rank, (uint32_t)world);
        const size_t count = (size_t)w * h * 4;
        float* d_sum = nullptr;
        HIPCHECK(hipMalloc((void**)&d_sum, count * sizeof(float)));
        hipStream_t stream;
        HIPCHECK(hipStreamCreate(&stream));
        // all ranks start the render together; its time is the slowest rank's (max-reduced below)
        float sync = 0.0f, *d_sync = nullptr;
        HIPCHECK(hipMalloc((void**)&d_sync, sizeof(float)));
        HIPCHECK(hipMemset(d_sync, 0, sizeof(float)));
        NCCLCHECK(ncclAllReduce(d_sync, d_sync, 1, ncclFloat, ncclSum, comm, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        const auto t1 = clock::now();
        if (range.second > 0) scene.render_range_dev(scene.camera, range.first, range.second, samples, seed, PT_RENDER_SUM, d_sum);
        else HIPCHECK(hipMemset(d_sum, 0, count * sizeof(float)));          // more ranks than samples: this rank adds nothing
        const auto t2 = clock::now();
        const pt_stats st = dev.stats();
        NCCLCHECK(ncclAllReduce(d_sum, d_sum, count, ncclFloat, ncclSum, comm, stream));      // RCCL over xGMI
        HIPCHECK(hipStreamSynchronize(stream));
        const auto t3 = clock::now();
        float render_ms = (float)ms(t1, t2);
        HIPCHECK(hipMemcpy(d_sync, &render_ms, sizeof(float), hipMemcpyHostToDevice));
        NCCLCHECK(ncclAllReduce(d_sync, d_sync, 1, ncclFloat, ncclMax, comm, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        HIPCHECK(hipMemcpy(&sync, d_sync, sizeof(float), hipMemcpyDeviceToHost));
        if (rank == 0) {
            std::vector<float> xyzw(count);
            HIPCHECK(hipMemcpy(xyzw.data(), d_sum, count * sizeof(float), hipMemcpyDeviceToHost));
            finalize_sums(xyzw);
            write_pfm(out + ".pfm", xyzw, w, h);
            std::printf("{\"program\": \"ptrender_mgpu\", \"gpus\": %d, \"width\": %u, \"height\": %u, \"spp\": %u, \"spp_rank0\": %u, "
                        "\"load_ms\": %.1f, \"render_s_slowest_rank\": %.4f, \"spp_per_s\": %.2f, \"allreduce_ms\": %.3f, "
                        "\"allreduce_bytes\": %zu, \"rank0_trace_Mrays_per_s\": %.0f, \"out\": \"%s.pfm\"}\n",
                        world, w, h, samples, range.second, ms(t0, t1), sync / 1e3, samples / (sync / 1e3 + ms(t2, t3) / 1e3), ms(t2, t3),
                        count * sizeof(float), st.trace_rays / std::max(st.trace_ms, 1e-9) / 1e3, out.c_str());
        }
        (void)hipFree(d_sum); (void)hipFree(d_sync); (void)hipStreamDestroy(stream);
    } catch (const pathtracer::Error& e) {
        std::fprintf(stderr, "ptrender_mgpu rank %d: error %d: %s\n", rank, e.code, e.what());
        ncclCommDestroy(comm);
        return 1;
    }
    NCCLCHECK(ncclCommDestroy(comm));
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: ptrender_mgpu scene.toml out_prefix [--gpus G] [--spp N] [--seed S]\n");
        return 2;
    }
    uint64_t seed = 1;
    long spp = -1;
    int gpus = 1;
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--seed") seed = std::strtoull(argv[i + 1], nullptr, 10);
        else if (k == "--spp") spp = std::strtol(argv[i + 1], nullptr, 10);
        else if (k == "--gpus") gpus = std::atoi(argv[i + 1]);
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (gpus < 1 || gpus > 64) { std::fprintf(stderr, "--gpus must be 1..64\n"); return 2; }
    const std::string id_file = std::string(argv[2]) + ".nccl_id." + std::to_string((long)getpid());
    std::remove(id_file.c_str());
    // fork the ranks before anything in this process touches the GPU
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; ++r) {
        const pid_t p = fork();
        if (p < 0) { std::perror("fork"); return 1; }
        if (p == 0) { const int rc = run_rank(r, gpus, argv[1], argv[2], spp, seed, id_file); std::fflush(nullptr); _exit(rc); }
        kids.push_back(p);
    }
    // a rank that fails (no such GPU, bad scene) leaves the others waiting in the communicator: end them with it
    int rc = 0;
    for (size_t left = kids.size(); left > 0; --left) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) { rc = 1; break; }
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            if (rc == 0) for (pid_t q : kids) if (q != p) kill(q, SIGTERM);
            rc = 1;
        }
    }
    std::remove(id_file.c_str());
    return rc;
}
