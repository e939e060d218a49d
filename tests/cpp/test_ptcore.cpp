// test_ptcore.cpp — the C++ host side (include/ptcore.hpp) exercised the way the reference tests
// itself, plus parity against the CPU oracle (oracle/pt_oracle.cpp: test infrastructure, linked here
// only as the checker).
//
//   triangle.rs:130-216   intersection, miss_t_negative, miss_u_plus_v_greater_than_one,
//                         miss_u_negative, miss_v_negative           -> through the GPU traversal
//   camera.rs:47-66       corners (up to normalisation: the reference's own test is stale)
//   bvh.rs:346-356        BVH::build against the oracle's recursive restatement, node for node
//   scene.rs:333 / main.rs:58-116   traverse and render against the oracle, bit for bit
//
//   test_ptcore --host    the cases that need no GPU
//   test_ptcore           everything (needs an MI355X)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/ptcore.hpp"

struct pto_scene;
extern "C" {
int pto_scene_create(const pt_scene_desc* desc, pto_scene** out);
void pto_scene_destroy(pto_scene* s);
uint32_t pto_scene_n_nodes(const pto_scene* s);
void pto_scene_get_nodes(const pto_scene* s, pt_bvh_node* out);
int pto_triangle_intersect(const float a[3], const float b[3], const float c[3], const float o[3], const float d[3],
                           float max_distance, float out[6]);
void pto_camera_ray(uint32_t w, uint32_t h, float sensor_width, float focal, float x, float y, float dir[3]);
void pto_srgb_to_xyz(const float rgb[3], float out[3]);
void pto_trace(const pto_scene* sc, const pt_ray* rays, const uint32_t* start, uint64_t n, pt_hit* out,
               uint64_t* counters, int nthreads);
void pto_render(const pto_scene* sc, const pt_camera* cam, uint32_t spp_begin, uint32_t spp_count, uint32_t spp_total,
                uint64_t seed, uint32_t mode, float* xyzw, uint64_t* counters, int nthreads);
}

using namespace pathtracer;

static int g_failed = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failed; } \
    } while (0)

static std::vector<Triangle> soup(uint32_t n) {
    std::vector<Triangle> t(n);
    const float lo[3] = {-1, -1, -4}, hi[3] = {1, 1, -2};
    if (pt_gen_soup(0x50554F53ull, n, lo, hi, 2.0f / std::cbrt((float)n), 0.0f, 8, t.data()) != PT_OK) std::abort();
    return t;
}
static std::vector<Material> soup_materials() {
    std::vector<Material> m(9, default_material());
    for (int i = 1; i < 9; ++i) { m[i].diffuse.value[0] = 0.1f * i; m[i].diffuse.value[1] = 0.9f - 0.1f * i; m[i].diffuse.value[2] = 0.5f; }
    m[8].emit_kind = PT_EMIT_ABSORBED_D65;
    m[8].absorption.value[0] = m[8].absorption.value[1] = m[8].absorption.value[2] = 1.0f;
    return m;
}
static pt_scene_desc desc_of(const std::vector<Triangle>& t, const BVH& b, const std::vector<Material>& m, const Sky& s) {
    pt_scene_desc d{};
    d.triangles = t.data(); d.n_triangles = (uint32_t)t.size();
    d.nodes = b.nodes.data(); d.n_nodes = (uint32_t)b.nodes.size();
    d.materials = m.data(); d.n_materials = (uint32_t)m.size();
    d.sky = s.sky;
    return d;
}
static bool same_bits(const void* a, const void* b, size_t n) { return std::memcmp(a, b, n) == 0; }
// NaN == NaN for images: the payload of an invalid-operation NaN differs between x86 and the GPU
static size_t differing_floats(const std::vector<float>& a, const std::vector<float>& b) {
    size_t n = a.size() != b.size();
    for (size_t i = 0; i < a.size() && i < b.size(); ++i)
        if (std::memcmp(&a[i], &b[i], 4) != 0 && !(std::isnan(a[i]) && std::isnan(b[i]))) ++n;
    return n;
}

// ---- host-only cases -------------------------------------------------------------------------------------

static void camera_corners() {       // camera.rs:47-66
    std::printf("camera::corners\n");
    // generate_ray is only exposed at pixel centres; the corner rays of the reference's test are the linear
    // extrapolation of the centre grid by half a pixel, checked here up to scale (directions are un-normalised)
    Camera cam;
    const std::vector<Ray> r = cam.generate_rays();
    const uint32_t W = cam.resolution[0], H = cam.resolution[1];
    const Ray &r00 = r[0], &r10 = r[1], &r01 = r[W];
    const float dx = r10.d[0] - r00.d[0], dy = r01.d[1] - r00.d[1];
    const float sx = 36.0f / (float)(W + 1), sy = 27.0f / (float)(H + 1);
    CHECK(std::fabs(dx - sx) < 1e-6f && std::fabs(dy + sy) < 1e-6f);
    CHECK(r00.d[2] == -50.0f && r00.o[0] == 0.0f && r00.o[1] == 0.0f && r00.o[2] == 0.0f);
    // centre of the frame looks straight down -Z
    const Ray& mid0 = r[(size_t)(H / 2) * W + W / 2];
    CHECK(std::fabs(mid0.d[0]) <= sx && std::fabs(mid0.d[1]) <= sy);
}

// the reference's ten unit tests, verbatim, against the CPU oracle: these pin the checker itself
static void reference_kats_on_the_oracle() {
    std::printf("oracle: triangle / camera / color KATs\n");
    const float a[3] = {0, 0, -1}, b[3] = {1, 0, -1}, c[3] = {0, 1, -1}, down[3] = {0, 0, -1}, up[3] = {0, 0, 1};
    float out[6];
    {   // triangle.rs:130-148 intersection
        const float o[3] = {1.0f / 3.0f, 1.0f / 3.0f, 0.0f};
        CHECK(pto_triangle_intersect(a, b, c, o, down, INFINITY, out) == 1 && out[0] == 1.0f && out[1] == 1.0f / 3.0f && out[2] == 1.0f / 3.0f);
    }
    {   // :150-165 miss_t_negative
        const float a2[3] = {0, 0, 2}, b2[3] = {1, 0, 2}, c2[3] = {0, 1, 2}, o[3] = {2.0f / 3.0f, 2.0f / 3.0f, 0.0f};
        CHECK(pto_triangle_intersect(a2, b2, c2, o, up, INFINITY, out) == 0);
    }
    const float o_uv[3] = {2.0f / 3.0f, 2.0f / 3.0f, 0.0f}, o_u[3] = {-2.0f / 3.0f, 2.0f / 3.0f, 0.0f}, o_v[3] = {2.0f / 3.0f, -2.0f / 3.0f, 0.0f};
    CHECK(pto_triangle_intersect(a, b, c, o_uv, up, INFINITY, out) == 0);      // :167-182 miss_u_plus_v_greater_than_one
    CHECK(pto_triangle_intersect(a, b, c, o_u, up, INFINITY, out) == 0);       // :184-199 miss_u_negative
    CHECK(pto_triangle_intersect(a, b, c, o_v, up, INFINITY, out) == 0);       // :201-216 miss_v_negative
    // camera.rs:47-66 corners: the expected vectors are normalised, generate_ray's are not (the reference's own test is
    // stale): compared up to scale
    const float px[4][2] = {{-0.5f, -0.5f}, {1024.5f, -0.5f}, {-0.5f, 768.5f}, {1024.5f, 768.5f}};
    const float want[4][3] = {{-18, 13.5f, -50}, {18, 13.5f, -50}, {-18, -13.5f, -50}, {18, -13.5f, -50}};
    for (int k = 0; k < 4; ++k) {
        float d[3];
        pto_camera_ray(1024, 768, 36.0f, 50.0f, px[k][0], px[k][1], d);
        const float s = d[2] / want[k][2];
        CHECK(s > 0.0f && std::fabs(d[0] - s * want[k][0]) < 1e-4f && std::fabs(d[1] - s * want[k][1]) < 1e-4f);
    }
    // color.rs:184-218 srgb_primary_{r,g,b,w}: chromaticity and luminance within 1e-3
    const float prim[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 1}};
    const float xyY[4][3] = {{0.64f, 0.33f, 0.2126f}, {0.30f, 0.60f, 0.7152f}, {0.15f, 0.06f, 0.0722f}, {0.3127f, 0.3290f, 1.0f}};
    for (int k = 0; k < 4; ++k) {
        float xyz[3];
        pto_srgb_to_xyz(prim[k], xyz);
        const float sum = xyz[0] + xyz[1] + xyz[2];
        CHECK(std::fabs(xyz[0] / sum - xyY[k][0]) < 1e-3f && std::fabs(xyz[1] / sum - xyY[k][1]) < 1e-3f && std::fabs(xyz[1] - xyY[k][2]) < 1e-3f);
    }
}

static void bvh_build_matches_the_recursive_restatement() {      // bvh.rs:134-248, 346-356
    std::printf("bvh::build\n");
    for (uint32_t n : {1u, 2u, 3u, 17u, 1000u, 20000u}) {
        const std::vector<Triangle> t = soup(n);
        const BVH b = BVH::build(t);
        CHECK(b.nodes.size() == 2 * (size_t)n - 1);
        const std::vector<Material> m = soup_materials();
        pt_scene_desc d = desc_of(t, BVH(), m, Sky::d65(0.0f));      // no nodes: the oracle builds its own
        pto_scene* o = nullptr;
        CHECK(pto_scene_create(&d, &o) == PT_OK);
        std::vector<pt_bvh_node> want(pto_scene_n_nodes(o));
        pto_scene_get_nodes(o, want.data());
        CHECK(want.size() == b.nodes.size() && same_bits(want.data(), b.nodes.data(), want.size() * sizeof(pt_bvh_node)));
        pto_scene_destroy(o);
    }
    CHECK(BVH::build({}).nodes.empty());
}

static void sky_errors() {       // scene.rs:121
    std::printf("scene::sky\n");
    const float up[3] = {0, 1, 0}, down[3] = {0, -1, 0}, sun[3] = {0.3f, 0.8f, 0.5f};
    bool threw = false;
    try { Sky::hosek_wilkie(3.0f, 0.3f, up, down); } catch (const Error& e) { threw = e.code == PT_ERR_INVALID; }
    CHECK(threw);
    const Sky s = Sky::hosek_wilkie(3.0f, 0.3f, up, sun);
    CHECK(s.sky.kind == PT_SKY_HOSEK_WILKIE && std::fabs(s.sky.zenith[1] - 1.0f) < 1e-6f);
}

// ---- GPU cases ----------------------------------------------------------------------------------------------

static Hit one_triangle(Device& dev, const float a[3], const float b[3], const float c[3], const float o[3], const float d[3]) {
    const std::vector<Triangle> t = {triangle(a, b, c)};
    const Scene s(dev, t, BVH::build(t), {default_material()}, Sky::d65(0.0f));
    return s.traverse({ray(o, d)})[0];
}

static void triangle_tests(Device& dev) {       // triangle.rs:130-216, through Scene::traverse on the GPU
    std::printf("triangle::intersection / miss_*\n");
    const float a[3] = {0, 0, -1}, b[3] = {1, 0, -1}, c[3] = {0, 1, -1};
    const float down[3] = {0, 0, -1}, up[3] = {0, 0, 1};
    {   // intersection
        const float o[3] = {1.0f / 3.0f, 1.0f / 3.0f, 0.0f};
        const Hit h = one_triangle(dev, a, b, c, o, down);
        CHECK(h.tri == 0 && h.t == 1.0f && h.u == 1.0f / 3.0f && h.v == 1.0f / 3.0f);
    }
    {   // miss_t_negative
        const float a2[3] = {0, 0, 2}, b2[3] = {1, 0, 2}, c2[3] = {0, 1, 2}, o[3] = {2.0f / 3.0f, 2.0f / 3.0f, 0.0f};
        CHECK(one_triangle(dev, a2, b2, c2, o, up).tri == PT_NO_TRIANGLE);
    }
    {   // miss_u_plus_v_greater_than_one
        const float o[3] = {2.0f / 3.0f, 2.0f / 3.0f, 0.0f};
        CHECK(one_triangle(dev, a, b, c, o, up).tri == PT_NO_TRIANGLE);
    }
    {   // miss_u_negative
        const float o[3] = {-2.0f / 3.0f, 2.0f / 3.0f, 0.0f};
        CHECK(one_triangle(dev, a, b, c, o, up).tri == PT_NO_TRIANGLE);
    }
    {   // miss_v_negative
        const float o[3] = {2.0f / 3.0f, -2.0f / 3.0f, 0.0f};
        CHECK(one_triangle(dev, a, b, c, o, up).tri == PT_NO_TRIANGLE);
    }
}

static void traverse_and_render_match_the_oracle(Device& dev) {
    std::printf("scene::traverse / render vs oracle\n");
    const std::vector<Triangle> t = soup(20000);
    const BVH b = BVH::build(t);
    const std::vector<Material> m = soup_materials();
    const float zen[3] = {0, 1, 0}, sun[3] = {0.3f, 0.8f, 0.5f};
    const Sky sky = Sky::hosek_wilkie(3.0f, 0.3f, zen, sun);
    const Scene scene(dev, t, b, m, sky);
    pt_scene_desc d = desc_of(t, b, m, sky);
    pto_scene* o = nullptr;
    CHECK(pto_scene_create(&d, &o) == PT_OK);

    Camera cam;
    cam.resolution[0] = 160; cam.resolution[1] = 90;
    const std::vector<Ray> rays = cam.generate_rays();
    std::vector<Hit> want(rays.size());
    pto_trace(o, rays.data(), nullptr, rays.size(), want.data(), nullptr, 8);
    for (uint32_t mode : {PT_TRAVERSE_MIN_T, PT_TRAVERSE_REPLAY, PT_TRAVERSE_EXACT}) {
        dev.set_traversal(mode);
        const std::vector<Hit> got = scene.traverse(rays);
        CHECK(same_bits(got.data(), want.data(), got.size() * sizeof(Hit)));
    }
    dev.set_traversal(PT_TRAVERSE_DEFAULT);
    {   // the scene hands its tree back as `BVH.nodes`; a tree built on the device is one the recursion can walk
        const std::vector<pt_bvh_node> back = scene.bvh();
        CHECK(back.size() == b.nodes.size() && same_bits(back.data(), b.nodes.data(), back.size() * sizeof(pt_bvh_node)));
        const Scene on_device = Scene::with_device_bvh(dev, t, m, sky);
        BVH db;
        db.nodes = on_device.bvh();
        CHECK(db.nodes.size() == 2 * t.size() - 1);
        pt_scene_desc dd = desc_of(t, db, m, sky);
        pto_scene* od = nullptr;
        CHECK(pto_scene_create(&dd, &od) == PT_OK);
        std::vector<Hit> want_d(rays.size());
        pto_trace(od, rays.data(), nullptr, rays.size(), want_d.data(), nullptr, 8);
        dev.set_traversal(PT_TRAVERSE_EXACT);
        const std::vector<Hit> got_d = on_device.traverse(rays);
        dev.set_traversal(PT_TRAVERSE_DEFAULT);
        CHECK(same_bits(got_d.data(), want_d.data(), got_d.size() * sizeof(Hit)));
        pto_scene_destroy(od);
    }
    size_t n_hit = 0;
    for (const Hit& h : want) n_hit += h.tri != PT_NO_TRIANGLE;
    CHECK(n_hit > rays.size() / 4);
    {   // the same batch from pinned buffers
        PinnedBuffer<Ray> pr(rays.size());
        PinnedBuffer<Hit> ph(rays.size());
        std::memcpy(pr.data(), rays.data(), rays.size() * sizeof(Ray));
        scene.traverse(pr.data(), nullptr, pr.size(), ph.data());
        CHECK(same_bits(ph.data(), want.data(), want.size() * sizeof(Hit)));
    }
    // bounce rays leaving the hit points, with `start` = the triangle they leave (bvh.rs:329-335)
    std::vector<Ray> bounce;
    std::vector<uint32_t> start;
    for (size_t i = 0; i < rays.size(); ++i) {
        if (want[i].tri == PT_NO_TRIANGLE) continue;
        Ray r{};
        for (int k = 0; k < 3; ++k) r.o[k] = rays[i].o[k] + rays[i].d[k] * want[i].t;
        r.d[0] = 0.3f - 0.001f * (float)(i % 700); r.d[1] = 0.5f; r.d[2] = 0.25f - 0.002f * (float)(i % 300);
        bounce.push_back(r);
        start.push_back(want[i].tri);
    }
    std::vector<Hit> want_b(bounce.size());
    pto_trace(o, bounce.data(), start.data(), bounce.size(), want_b.data(), nullptr, 8);
    const std::vector<Hit> got_b = scene.traverse(bounce, &start);
    CHECK(same_bits(got_b.data(), want_b.data(), got_b.size() * sizeof(Hit)));

    // render: 80x45, 4 spp, seed 1; FINAL, and two SUM halves that add up to it
    cam.resolution[0] = 80; cam.resolution[1] = 45;
    const pt_camera pc = cam.ffi();
    std::vector<float> want_img((size_t)80 * 45 * 4);
    pto_render(o, &pc, 0, 4, 4, 1, PT_RENDER_FINAL, want_img.data(), nullptr, 8);
    const std::vector<float> got_img = scene.render(cam, 4, 1);
    CHECK(differing_floats(got_img, want_img) == 0);
    const std::vector<float> h0 = scene.render_range(cam, 0, 2, 4, 1), h1 = scene.render_range(cam, 2, 2, 4, 1);
    size_t bad = 0;
    for (size_t p = 0; p < (size_t)80 * 45; ++p) {
        CHECK(h0[4 * p + 3] == 2.0f && h1[4 * p + 3] == 2.0f);
        for (int c = 0; c < 3; ++c) {
            const float sum = (h0[4 * p + c] + h1[4 * p + c]) * 0.25f, ref = got_img[4 * p + c];
            if (std::isnan(sum) != std::isnan(ref) || (!std::isnan(sum) && std::fabs(sum - ref) > 1e-5f * std::fabs(ref) + 1e-12f)) ++bad;
        }
    }
    CHECK(bad == 0);
    // three "ranks" over 4 samples: ranges {0,2},{2,1},{3,1}; summed and finalised = the one-call image up to f32 order
    std::vector<float> total((size_t)80 * 45 * 4, 0.0f);
    uint32_t covered = 0;
    for (uint32_t rank = 0; rank < 3; ++rank) {
        const auto rg = sample_range(4, rank, 3);
        CHECK(rg.first == covered);
        covered += rg.second;
        const std::vector<float> part = scene.render_range(cam, rg.first, rg.second, 4, 1);
        for (size_t k = 0; k < total.size(); ++k) total[k] += part[k];
    }
    CHECK(covered == 4);
    finalize_sums(total);
    size_t off = 0;
    for (size_t k = 0; k < total.size(); ++k) {
        const float ref = got_img[k];
        if (std::isnan(total[k]) != std::isnan(ref) || (!std::isnan(ref) && std::fabs(total[k] - ref) > 1e-5f * std::fabs(ref) + 1e-12f)) ++off;
    }
    CHECK(off == 0);
    pto_scene_destroy(o);
}

static void load_and_render(Device& dev, const std::string& dir) {       // Scene::load (scene.rs:194-282)
    std::printf("scene::load\n");
    std::ofstream(dir + "/q.obj") << "mtllib q.mtl\nv -1 -1 -3\nv 1 -1 -3\nv 1 1 -3\nv -1 1 -3\nv -4 -1.5 0\nv 4 -1.5 0\nv 4 -1.5 -8\nv -4 -1.5 -8\n"
                                     "usemtl lamp\nf 1 2 3 4\nusemtl floor\nf 5 6 7 8\n";
    std::ofstream(dir + "/q.mtl") << "newmtl lamp\nKd 0.2 0.2 0.2\nKe 1.0 0.9 0.8\nnewmtl floor\nKd 0.7 0.6 0.5\n";
    std::ofstream(dir + "/q.toml") << "[camera]\nresolution = [48, 32]\nsamples = 4\n[sky]\npower = 0.5\n[[meshes]]\npath = \"q.obj\"\n";
    const Scene scene = Scene::load(dev, dir + "/q.toml");
    CHECK(scene.camera.resolution[0] == 48 && scene.camera.resolution[1] == 32 && scene.camera.samples == 4);
    CHECK(scene.info().n_triangles == 4);
    const std::vector<float> img = scene.render(7);

    pt_scene_file* f = nullptr;
    CHECK(pt_scene_file_load((dir + "/q.toml").c_str(), &f) == PT_OK);
    pt_scene_file_info info;
    pt_scene_file_get(f, &info);
    pto_scene* o = nullptr;
    CHECK(pto_scene_create(&info.desc, &o) == PT_OK);
    std::vector<float> want(img.size());
    pto_render(o, &info.camera, 0, 4, 4, 7, PT_RENDER_FINAL, want.data(), nullptr, 4);
    CHECK(differing_floats(img, want) == 0);
    float lum = 0.0f;
    for (size_t p = 0; p < img.size() / 4; ++p) lum += img[4 * p + 1];
    CHECK(lum > 0.0f);
    const Display disp = tonemap(dev, img, 48, 32);
    CHECK(disp.bgra8.size() == img.size() && disp.avg_luminance > 0.0f && disp.max_luminance >= disp.avg_luminance);
    pto_scene_destroy(o);
    pt_scene_file_free(f);

    bool threw = false;
    try { Scene::load(dev, dir + "/absent.toml"); } catch (const Error& e) { threw = e.code == PT_ERR_INVALID; }
    CHECK(threw);
}

int main(int argc, char** argv) {
    const bool host_only = argc > 1 && std::string(argv[1]) == "--host";
    const std::string tmp = argc > 2 ? argv[2] : "/tmp";
    camera_corners();
    reference_kats_on_the_oracle();
    bvh_build_matches_the_recursive_restatement();
    sky_errors();
    if (!host_only) {
        try {
            Device dev(0);
            triangle_tests(dev);
            traverse_and_render_match_the_oracle(dev);
            load_and_render(dev, tmp);
        } catch (const Error& e) {
            std::printf("  FAILED: pathtracer::Error %d: %s\n", e.code, e.what());
            ++g_failed;
        }
    }
    std::printf(g_failed ? "%d check(s) FAILED\n" : "all checks passed\n", g_failed);
    return g_failed ? 1 : 0;
}
