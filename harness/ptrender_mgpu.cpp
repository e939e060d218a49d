// ptrender_mgpu — the sample-split multi-GPU render of BASELINE configs[3] from a compiled host: one process per GPU, the
// scene replicated, rank g renders sample indices sample_range(spp, g, G) of every pixel into a PT_RENDER_SUM buffer in HBM,
// ONE ncclAllReduce(ncclFloat, ncclSum) over xGMI combines the W*H*4-float buffers, finalize_sums divides by the samples
// (main.rs:35-52,100-107 is the shared framebuffer this replaces).  The Rust host would do the same through
// the FFI crate of INTEGRATION.md; this is the C++ rank program the repository can build and test.
//
//   ptrender_mgpu scene.toml out_prefix [--gpus G] [--spp N] [--seed S] [--size WxH]
//   ptrender_mgpu --soup N   out_prefix [--gpus G] [--spp N] [--seed S] [--size WxH] [--sky hw|d65]
//     (BASELINE configs[3] = --soup 1000000 --sky hw --size 3840x2160 --spp 1024 --gpus 8)
//
// The launcher process never touches the GPU: it forks G rank processes BEFORE any HIP call (a forked child of a process
// that has initialised the GPU is not usable), hands them the path of a rendezvous file, and waits.  Rank 0 makes the
// ncclUniqueId and publishes it through that file (write + rename); every rank LOADS ITS SCENE FIRST and then joins with
// ncclCommInitRank; the first collective is an agreement (did every rank load? how many ranks are there?), so a bad scene ends
// all ranks by agreement, not by a signal.  Rank g uses GPU g.  Rank 0 writes out_prefix.pfm / .png and one JSON line with the timings (render seconds = slowest rank).
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
#include <cmath>
#include <fstream>
#include <optional>
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

struct Job {
    std::string scene_path;       // TOML scene (Scene::load), or empty with soup > 0
    std::string out;
    long soup = 0;                // --soup N: the synthetic soup of SURVEY 8(d) (configs 3/4), seed "SOUP"
    bool sky_hw = true;           // --sky hw | d65
    uint32_t width = 0, height = 0;   // --size WxH (0: the scene file's / 1920x1080 for the soup)
    long spp = -1;
    uint64_t seed = 1;
};

// soup-N as pathtracer_amd/workloads.py makes it (same generator, same materials, same sky): configs 3 and 4 from a compiled host
pathtracer::Scene make_soup(pathtracer::Device& dev, const Job& job) {
    using namespace pathtracer;
    const uint32_t n = (uint32_t)job.soup;
    std::vector<Triangle> tris(n);
    const float lo[3] = {-1, -1, -4}, hi[3] = {1, 1, -2};
    const float edge = (float)(2.0 / std::pow((double)n, 1.0 / 3.0));
    if (pt_gen_soup(0x50554F53ull, n, lo, hi, edge, 0.0f, 8, tris.data()) != PT_OK) throw Error(PT_ERR_INVALID, "pt_gen_soup");
    std::vector<Material> mats(9, default_material());
    const float col[7][3] = {{0.75f, 0.25f, 0.25f}, {0.25f, 0.75f, 0.25f}, {0.25f, 0.25f, 0.75f}, {0.75f, 0.75f, 0.25f},
                             {0.25f, 0.75f, 0.75f}, {0.75f, 0.25f, 0.75f}, {0.60f, 0.60f, 0.60f}};
    for (int i = 0; i < 7; ++i) for (int k = 0; k < 3; ++k) mats[1 + i].diffuse.value[k] = col[i][k];
    const float emit[3] = {1.0f, 0.9f, 0.7f};
    for (int k = 0; k < 3; ++k) { mats[8].diffuse.value[k] = 0.8f; mats[8].absorption.value[k] = emit[k]; }
    mats[8].emit_kind = PT_EMIT_ABSORBED_D65;
    const float zen[3] = {0, 1, 0}, sun[3] = {0.3f, 0.8f, 0.5f};
    const Sky sky = job.sky_hw ? Sky::hosek_wilkie(3.0f, 0.3f, zen, sun) : Sky::d65(0.0f);
    Camera cam;
    cam.resolution[0] = job.width ? job.width : 1920; cam.resolution[1] = job.height ? job.height : 1080;
    cam.samples = 64;
    const BVH bvh = BVH::build(tris);
    return Scene(dev, tris, bvh, mats, sky, {}, cam);
}

int run_rank(int rank, int world, const Job& job, const std::string& id_file) {
    g_rank = rank;
    using namespace pathtracer;
    using clock = std::chrono::steady_clock;
    const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    int n_dev = 0;
    HIPCHECK(hipGetDeviceCount(&n_dev));
    if (rank >= n_dev) { std::fprintf(stderr, "rank %d: only %d GPUs visible (one rank per GPU)\n", rank, n_dev); return 1; }
    HIPCHECK(hipSetDevice(rank));
    // ---- the scene first: a rank that cannot load it says so through the communicator instead of leaving its peers in it ----
    const auto t0 = clock::now();
    std::optional<Device> dev;
    std::optional<Scene> scene;
    float failed = 0.0f;
    try {
        dev.emplace(rank);
        if (job.soup > 0) scene.emplace(make_soup(*dev, job));
        else scene.emplace(Scene::load(*dev, job.scene_path));
        if (job.width) { scene->camera.resolution[0] = job.width; scene->camera.resolution[1] = job.height; }
    } catch (const pathtracer::Error& e) {
        std::fprintf(stderr, "ptrender_mgpu rank %d: error %d: %s\n", rank, e.code, e.what());
        failed = 1.0f;
    }
    const auto t_loaded = clock::now();
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
            if (tries > 60000) { std::fprintf(stderr, "rank %d: no rendezvous file after 600 s\n", rank); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    ncclComm_t comm;
    NCCLCHECK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));
    float sync = 0.0f, *d_sync = nullptr;
    HIPCHECK(hipMalloc((void**)&d_sync, 2 * sizeof(float)));
    // agreement: {any rank failed to load, ranks present}; also the barrier all ranks start the render from
    const float hello[2] = {failed, 1.0f};
    float agreed[2] = {0, 0};
    HIPCHECK(hipMemcpy(d_sync, hello, sizeof(hello), hipMemcpyHostToDevice));
    NCCLCHECK(ncclAllReduce(d_sync, d_sync, 2, ncclFloat, ncclSum, comm, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    HIPCHECK(hipMemcpy(agreed, d_sync, sizeof(agreed), hipMemcpyDeviceToHost));
    const int ranks_seen = (int)agreed[1];
    if (agreed[0] != 0.0f || ranks_seen != world) {
        if (rank == 0) std::fprintf(stderr, "ptrender_mgpu: %d of %d ranks could not load the scene (%d ranks in the communicator): nothing rendered\n",
                                    (int)agreed[0], world, ranks_seen);
        (void)hipFree(d_sync); (void)hipStreamDestroy(stream);
        ncclCommDestroy(comm);
        return 1;
    }
    try {
        const uint32_t w = scene->camera.resolution[0], h = scene->camera.resolution[1];
        const uint32_t samples = job.spp > 0 ? (uint32_t)job.spp : scene->camera.samples;
        const auto range = sample_range(samples, (uint32_t)rank, (uint32_t)world);
        const size_t count = (size_t)w * h * 4;
        float* d_sum = nullptr;
        HIPCHECK(hipMalloc((void**)&d_sum, count * sizeof(float)));
        const auto t1 = clock::now();
        if (range.second > 0) scene->render_range_dev(scene->camera, range.first, range.second, samples, job.seed, PT_RENDER_SUM, d_sum);
        else HIPCHECK(hipMemset(d_sum, 0, count * sizeof(float)));          // more ranks than samples: this rank adds nothing
        const auto t2 = clock::now();
        const pt_stats st = dev->stats();
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
            write_pfm(job.out + ".pfm", xyzw, w, h);
            std::printf("{\"program\": \"ptrender_mgpu\", \"gpus\": %d, \"ranks_seen\": %d, \"workload\": \"%s\", \"width\": %u, \"height\": %u, \"spp\": %u, \"spp_rank0\": %u, "
                        "\"load_ms\": %.1f, \"render_s_slowest_rank\": %.4f, \"spp_per_s\": %.2f, \"allreduce_ms\": %.3f, "
                        "\"allreduce_bytes\": %zu, \"rank0_trace_Mrays_per_s\": %.0f, \"out\": \"%s.pfm\"}\n",
                        world, ranks_seen, job.soup > 0 ? (std::string("soup-") + std::to_string(job.soup) + (job.sky_hw ? " + Hosek-Wilkie sky" : " + D65 sky")).c_str() : job.scene_path.c_str(),
                        w, h, samples, range.second, ms(t0, t_loaded), sync / 1e3, samples / (sync / 1e3 + ms(t2, t3) / 1e3), ms(t2, t3),
                        count * sizeof(float), st.trace_rays / std::max(st.trace_ms, 1e-9) / 1e3, job.out.c_str());
        }
        (void)hipFree(d_sum); (void)hipFree(d_sync); (void)hipStreamDestroy(stream);
    } catch (const pathtracer::Error& e) {
        // a render error after the agreement: peers may sit in the all-reduce; the parent ends them (see main)
        std::fprintf(stderr, "ptrender_mgpu rank %d: error %d: %s\n", rank, e.code, e.what());
        ncclCommDestroy(comm);
        return 1;
    }
    scene.reset();
    dev.reset();
    NCCLCHECK(ncclCommDestroy(comm));
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    const char* usage = "usage: ptrender_mgpu (scene.toml | --soup N) out_prefix [--gpus G] [--spp N] [--seed S] [--size WxH] [--sky hw|d65]\n";
    Job job;
    int gpus = 1;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        if (k.rfind("--", 0) != 0) { pos.push_back(k); continue; }
        if (i + 1 >= argc) { std::fprintf(stderr, "%s", usage); return 2; }
        const char* v = argv[++i];
        if (k == "--seed") job.seed = std::strtoull(v, nullptr, 10);
        else if (k == "--spp") job.spp = std::strtol(v, nullptr, 10);
        else if (k == "--gpus") gpus = std::atoi(v);
        else if (k == "--soup") job.soup = std::strtol(v, nullptr, 10);
        else if (k == "--sky") {
            if (std::strcmp(v, "hw") == 0) job.sky_hw = true;
            else if (std::strcmp(v, "d65") == 0) job.sky_hw = false;
            else { std::fprintf(stderr, "--sky must be hw or d65\n"); return 2; }
        } else if (k == "--size") {
            unsigned w = 0, h = 0;
            if (std::sscanf(v, "%ux%u", &w, &h) != 2 || w == 0 || h == 0 || w > 65536 || h > 65536) { std::fprintf(stderr, "--size must be WxH\n"); return 2; }
            job.width = w; job.height = h;
        } else { std::fprintf(stderr, "unknown option %s\n%s", k.c_str(), usage); return 2; }
    }
    if (job.soup < 0 || job.soup > 200000000) { std::fprintf(stderr, "--soup must be 1..200000000\n"); return 2; }
    if (job.soup > 0 && pos.size() == 1) job.out = pos[0];
    else if (job.soup == 0 && pos.size() == 2) { job.scene_path = pos[0]; job.out = pos[1]; }
    else { std::fprintf(stderr, "%s", usage); return 2; }
    if (gpus < 1 || gpus > 64) { std::fprintf(stderr, "--gpus must be 1..64\n"); return 2; }
    const std::string id_file = job.out + ".nccl_id." + std::to_string((long)getpid());
    std::remove(id_file.c_str());
    // fork the ranks before anything in this process touches the GPU
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; ++r) {
        const pid_t p = fork();
        if (p < 0) { std::perror("fork"); return 1; }
        if (p == 0) { const int rc = run_rank(r, gpus, job, id_file); std::fflush(nullptr); _exit(rc); }
        kids.push_back(p);
    }
    // Ranks that cannot load the scene agree on that through the communicator and all leave.  A rank that cannot even join
    // (no such GPU) or fails later leaves the others waiting in a collective: those are ended here.
    int rc = 0;
    for (size_t left = kids.size(); left > 0; --left) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) { rc = 1; break; }
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            if (rc == 0 && left > 1) {
                // give peers that are leaving by agreement a moment before ending the rest
                std::this_thread::sleep_for(std::chrono::milliseconds(500));
                for (pid_t q : kids) if (q != p) kill(q, SIGTERM);
            }
            rc = 1;
        }
    }
    std::remove(id_file.c_str());
    return rc;
}
