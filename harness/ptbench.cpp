// ptbench — the headline workloads without Python: BASELINE configs[1] (closest hit of the 1080p pixel-centre rays
// against soup-1M) through the host-pointer entry point from pinned and pageable buffers, and configs[2]
// (spectral path tracing under the Hosek-Wilkie sky).  bench.py is the driver's benchmark (device-resident
// buffers, roofline, CPU baseline); this program shows what a C++ or Rust host sees through include/ptcore.hpp.
//   ptbench [--tris N] [--spp N] [--reps N] [--lbvh]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/ptcore.hpp"

using namespace pathtracer;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); }

int main(int argc, char** argv) {
    uint32_t n_tris = 1000000, spp = 64, reps = 10;
    bool lbvh = false;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        if (k == "--lbvh") lbvh = true;
        else if (i + 1 < argc && k == "--tris") n_tris = (uint32_t)std::atol(argv[++i]);
        else if (i + 1 < argc && k == "--spp") spp = (uint32_t)std::atol(argv[++i]);
        else if (i + 1 < argc && k == "--reps") reps = (uint32_t)std::atol(argv[++i]);
        else { std::fprintf(stderr, "usage: ptbench [--tris N] [--spp N] [--reps N] [--lbvh]\n"); return 2; }
    }
    try {
        Device dev(0);
        // soup scene of SURVEY.md 8(d): centres in [-1,1]^2 x [-4,-2], edge scale = cube side / N^(1/3)
        std::vector<Triangle> tris(n_tris);
        const float lo[3] = {-1, -1, -4}, hi[3] = {1, 1, -2};
        if (pt_gen_soup(0x50554F53ull, n_tris, lo, hi, 2.0f / std::cbrt((float)n_tris), 0.0f, 8, tris.data()) != PT_OK) return 1;
        std::vector<Material> mats(9, default_material());
        const float col[7][3] = {{0.8f, 0.2f, 0.2f}, {0.2f, 0.8f, 0.2f}, {0.2f, 0.2f, 0.8f}, {0.8f, 0.8f, 0.2f}, {0.8f, 0.2f, 0.8f}, {0.2f, 0.8f, 0.8f}, {0.7f, 0.7f, 0.7f}};
        for (int i = 0; i < 7; ++i) for (int k = 0; k < 3; ++k) mats[1 + i].diffuse.value[k] = col[i][k];
        mats[8].emit_kind = PT_EMIT_ABSORBED_D65;
        for (int k = 0; k < 3; ++k) mats[8].absorption.value[k] = 1.0f;
        const float zen[3] = {0, 1, 0}, sun[3] = {0.3f, 0.8f, 0.5f};
        const Sky sky = Sky::hosek_wilkie(3.0f, 0.3f, zen, sun);
        auto t = Clock::now();
        BVH bvh;
        if (!lbvh) bvh = BVH::build(tris);
        const double build_ms = ms_since(t);
        t = Clock::now();
        const Scene scene = lbvh ? Scene::with_device_bvh(dev, tris, mats, sky) : Scene(dev, tris, bvh, mats, sky);
        std::printf("scene: %u triangles, %s %.1f ms, upload %.1f ms, depth %u, %.0f MB in HBM\n", n_tris,
                    lbvh ? "device BVH build inside upload," : "host BVH::build", build_ms, ms_since(t), scene.info().max_depth,
                    scene.info().device_bytes / 1e6);

        Camera cam;
        cam.resolution[0] = 1920; cam.resolution[1] = 1080;
        const std::vector<Ray> rays = cam.generate_rays();
        const size_t n = rays.size();
        std::vector<Hit> hits(n);
        scene.traverse(rays.data(), nullptr, n, hits.data());
        t = Clock::now();
        for (uint32_t r = 0; r < reps; ++r) scene.traverse(rays.data(), nullptr, n, hits.data());
        double per = ms_since(t) / reps;
        std::printf("closest hit, %zu rays, pageable host buffers: %.2f ms per batch = %.0f Mrays/s (PCIe-inclusive)\n", n, per, n / per / 1e3);
        PinnedBuffer<Ray> pr(n);
        PinnedBuffer<Hit> ph(n);
        std::memcpy(pr.data(), rays.data(), n * sizeof(Ray));
        for (int r = 0; r < 6; ++r) scene.traverse(pr.data(), nullptr, n, ph.data());
        t = Clock::now();
        for (uint32_t r = 0; r < reps; ++r) scene.traverse(pr.data(), nullptr, n, ph.data());
        per = ms_since(t) / reps;
        const pt_stats st = dev.stats();
        std::printf("closest hit, %zu rays, pinned host buffers:   %.2f ms per batch = %.0f Mrays/s (PCIe-inclusive); kernels alone %.2f ms = %.0f Mrays/s\n",
                    n, per, n / per / 1e3, st.trace_ms, n / st.trace_ms / 1e3);
        size_t n_hit = 0;
        for (size_t i = 0; i < n; ++i) n_hit += ph[i].tri != PT_NO_TRIANGLE;
        std::printf("hit fraction %.4f, same hits from both buffers: %s\n", (double)n_hit / n,
                    std::memcmp(ph.data(), hits.data(), n * sizeof(Hit)) == 0 ? "yes" : "NO");

        if (spp) {
            scene.render(cam, spp, 1);                      // first call sizes the workspace
            t = Clock::now();
            const std::vector<float> img = scene.render(cam, spp, 1);
            const double s = ms_since(t) / 1e3;
            const pt_stats rs = dev.stats();
            std::printf("spectral, 1920x1080, %u spp: %.3f s = %.1f spp/s (device %.1f ms, %.2f segments per path, traversal %.0f Mrays/s)\n",
                        spp, s, spp / s, rs.total_ms, (double)rs.segments / (double)rs.paths, rs.trace_rays / rs.trace_ms / 1e3);
        }
    } catch (const Error& e) {
        std::fprintf(stderr, "ptbench: error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
