// ptcore.hpp — C++17 host side over the C ABI of ptcore.h (header-only, no HIP types).
//
// The reference is one Rust crate; a Rust host would bind ptcore.h through a `ptcore-sys` crate
// (INTEGRATION.md).  Rust is not available in this image, so the host side above the C ABI is written
// in C++ and mirrors the reference's own types and call sites, name for name where C++ allows:
//
//   pathtracer::BVH::build(shapes)                 BVH::build                (bvh.rs:346-356)
//   pathtracer::Camera::generate_ray(x, y)         Camera::generate_ray      (camera.rs:19-28)
//   pathtracer::Sky::d65 / Sky::hosek_wilkie       InitialisedSky::from_sky  (scene.rs:105-129)
//   pathtracer::Scene::load(device, path)          Scene::load               (scene.rs:194-282)
//   scene.traverse(rays, start)                    self.bvh.traverse(..)     (scene.rs:333), batched
//   scene.render(camera, samples, seed)            the render closure        (main.rs:58-116)
//   pathtracer::tonemap(device, xyzw, w, h)        renderer.rs:297-309 + shader.wgsl:32-44
//
// Error behaviour: where the reference returns Err or panics (malformed scene, bad sky, degenerate
// BVH) these throw pathtracer::Error carrying the C status code and the library's error text.
// Everything that computes on rays or samples runs on the GPU through libptcore.so; there is no CPU
// path here.
#ifndef PTCORE_HPP
#define PTCORE_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ptcore.h"

namespace pathtracer {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

using Triangle = pt_triangle;     // Triangle::new arguments (triangle.rs:42-49)
using Material = pt_material;     // material.rs:198-203
using Ray = pt_ray;               // bvh.rs:372-387
using Hit = pt_hit;               // Option<(&Triangle, Intersection)>: tri == PT_NO_TRIANGLE is None

inline Triangle triangle(const float a[3], const float b[3], const float c[3], uint32_t material_index = 0) {
    Triangle t{};
    for (int k = 0; k < 3; ++k) { t.a[k] = a[k]; t.b[k] = b[k]; t.c[k] = c[k]; }
    t.material_index = material_index;
    return t;
}
inline Ray ray(const float origin[3], const float direction[3]) {      // Ray::new: direction kept as given
    Ray r{};
    for (int k = 0; k < 3; ++k) { r.o[k] = origin[k]; r.d[k] = direction[k]; }
    return r;
}
inline Material default_material() { Material m; pt_material_default(&m); return m; }   // Material::default()

// BVH (bvh.rs:341-343): the reference-order node vector.
struct BVH {
    std::vector<pt_bvh_node> nodes;
    // BVH::build (bvh.rs:346-356): bit-faithful binned-SAH build on the host
    static BVH build(const std::vector<Triangle>& shapes) {
        BVH b;
        if (shapes.empty()) return b;
        b.nodes.resize(2 * shapes.size() - 1);
        uint32_t n = 0;
        const int rc = pt_bvh_build(shapes.data(), (uint32_t)shapes.size(), b.nodes.data(), &n);
        if (rc != PT_OK) throw Error(rc, rc == PT_ERR_BVH_DEGENERATE ? "BVH::build: empty child box (bvh.rs:242)" : "BVH::build failed");
        b.nodes.resize(n);
        return b;
    }
};

// scene::Camera (scene.rs:134-152) + Camera (camera.rs:11-39)
struct Camera {
    uint32_t resolution[2] = {1024, 768};
    float sensor_width = 36.0f, focal_length = 50.0f;
    uint32_t samples = 128;
    pt_camera ffi() const { return pt_camera{resolution[0], resolution[1], sensor_width, focal_length}; }
    // every pixel-centre ray, row-major: generate_ray(x, y) for y in 0..H, x in 0..W
    std::vector<Ray> generate_rays() const {
        std::vector<Ray> rays((size_t)resolution[0] * resolution[1]);
        const pt_camera c = ffi();
        if (!rays.empty() && pt_gen_primary_rays(&c, rays.data()) != PT_OK) throw Error(PT_ERR_INVALID, "Camera::generate_rays");
        return rays;
    }
};

struct Sky {
    pt_sky sky{};
    static Sky d65(float power) { Sky s; pt_sky_d65(power, &s.sky); return s; }       // Sky::D65
    // Sky::HosekWilkie; camera_transform column-major (nullptr = identity); throws where scene.rs:121 asserts
    static Sky hosek_wilkie(float turbidity, float albedo, const float zenith[3], const float sun[3],
                            const float* camera_transform = nullptr) {
        Sky s;
        const int rc = pt_sky_hosek_wilkie(turbidity, albedo, zenith, sun, camera_transform, &s.sky);
        if (rc != PT_OK) throw Error(rc, "Sky::hosek_wilkie: solar elevation must be > 0 and turbidity within [1, 10]");
        return s;
    }
};

// Pinned host memory (pt_host_alloc): ray / hit buffers kept in one make Scene::traverse run at PCIe speed.
template <class T>
class PinnedBuffer {
public:
    explicit PinnedBuffer(size_t count) : n_(count) {
        void* p = nullptr;
        const int rc = pt_host_alloc(count * sizeof(T), &p);
        if (rc != PT_OK) throw Error(rc, pt_last_error(nullptr));
        p_ = static_cast<T*>(p);
    }
    ~PinnedBuffer() { pt_host_free(p_); }
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    T* data() { return p_; }
    const T* data() const { return p_; }
    size_t size() const { return n_; }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }

private:
    T* p_ = nullptr;
    size_t n_ = 0;
};

// One GPU (pt_ctx).  One caller thread per Device.
class Device {
public:
    explicit Device(int device_id = 0) {
        const int rc = pt_init(&device_id, 1, &ctx_);
        if (rc != PT_OK) throw Error(rc, pt_last_error(nullptr));
    }
    ~Device() { pt_shutdown(ctx_); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    pt_ctx* handle() const { return ctx_; }
    void set_traversal(uint32_t mode) { check(pt_set_traversal(ctx_, mode)); }
    pt_stats stats() const { pt_stats s{}; pt_get_stats(ctx_, &s); return s; }
    void check(int rc) const { if (rc != PT_OK) throw Error(rc, pt_last_error(ctx_)); }

private:
    pt_ctx* ctx_ = nullptr;
};

// Scene { bvh, triangles, materials, sky, camera } (scene.rs:185-191), resident in HBM.
class Scene {
public:
    Camera camera;

    Scene(Device& dev, const std::vector<Triangle>& triangles, const BVH& bvh, const std::vector<Material>& materials,
          const Sky& sky, const std::vector<pt_image>& images = {}, const Camera& cam = Camera())
        : camera(cam), dev_(&dev) {
        pt_scene_desc d{};
        d.triangles = triangles.data(); d.n_triangles = (uint32_t)triangles.size();
        d.nodes = bvh.nodes.data(); d.n_nodes = (uint32_t)bvh.nodes.size();
        d.materials = materials.data(); d.n_materials = (uint32_t)materials.size();
        d.images = images.data(); d.n_images = (uint32_t)images.size();
        d.sky = sky.sky;
        dev.check(pt_scene_create(dev.handle(), &d, &scene_));
    }
    // the same scene with the BVH built on the GPU (LBVH, pt_scene_create_ex; no counterpart in the reference)
    static Scene with_device_bvh(Device& dev, const std::vector<Triangle>& triangles, const std::vector<Material>& materials,
                                 const Sky& sky, const std::vector<pt_image>& images = {}, const Camera& cam = Camera()) {
        Scene s(dev);
        s.camera = cam;
        pt_scene_desc d{};
        d.triangles = triangles.data(); d.n_triangles = (uint32_t)triangles.size();
        d.materials = materials.data(); d.n_materials = (uint32_t)materials.size();
        d.images = images.data(); d.n_images = (uint32_t)images.size();
        d.sky = sky.sky;
        dev.check(pt_scene_create_ex(dev.handle(), &d, PT_SCENE_BUILD_LBVH, &s.scene_));
        return s;
    }
    // Scene::load (scene.rs:194-282): TOML scene file + OBJ/MTL + PNG, BVH::build, upload
    static Scene load(Device& dev, const std::string& path) {
        pt_scene_file* f = nullptr;
        int rc = pt_scene_file_load(path.c_str(), &f);
        if (rc != PT_OK) throw Error(rc, pt_last_error(nullptr));
        pt_scene_file_info info;
        pt_scene_file_get(f, &info);
        Scene s(dev);
        s.camera.resolution[0] = info.camera.width; s.camera.resolution[1] = info.camera.height;
        s.camera.sensor_width = info.camera.sensor_width; s.camera.focal_length = info.camera.focal_length;
        s.camera.samples = info.samples;
        rc = pt_scene_create(dev.handle(), &info.desc, &s.scene_);
        pt_scene_file_free(f);
        dev.check(rc);
        return s;
    }
    ~Scene() { if (scene_) pt_scene_destroy(dev_->handle(), scene_); }
    Scene(Scene&& o) noexcept : camera(o.camera), dev_(o.dev_), scene_(o.scene_) { o.scene_ = nullptr; }
    Scene& operator=(Scene&& o) noexcept {
        if (this != &o) { if (scene_) pt_scene_destroy(dev_->handle(), scene_); camera = o.camera; dev_ = o.dev_; scene_ = o.scene_; o.scene_ = nullptr; }
        return *this;
    }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    const pt_scene* handle() const { return scene_; }
    pt_scene_info info() const { pt_scene_info i{}; pt_scene_get_info(scene_, &i); return i; }
    // `BVH { nodes }` of this scene (bvh.rs:341-343), also for a tree built on the device
    std::vector<pt_bvh_node> bvh() const {
        uint32_t n = 0;
        dev_->check(pt_scene_get_bvh(dev_->handle(), scene_, nullptr, 0, &n));
        std::vector<pt_bvh_node> nodes(n);
        dev_->check(pt_scene_get_bvh(dev_->handle(), scene_, nodes.data(), n, &n));
        return nodes;
    }

    // self.bvh.traverse(&ray, start, &self.triangles) (scene.rs:333) for a batch; start[i] == PT_NO_TRIANGLE is None
    std::vector<Hit> traverse(const std::vector<Ray>& rays, const std::vector<uint32_t>* start = nullptr) const {
        if (start && start->size() != rays.size()) throw Error(PT_ERR_INVALID, "Scene::traverse: start has the wrong length");
        std::vector<Hit> hits(rays.size());
        dev_->check(pt_trace_closest(dev_->handle(), scene_, rays.data(), start ? start->data() : nullptr, (uint32_t)rays.size(), hits.data()));
        return hits;
    }
    // the same over caller-owned buffers (pageable or pinned); start may be nullptr
    void traverse(const Ray* rays, const uint32_t* start, size_t n, Hit* hits) const {
        if (n > 0xF0000000ull) throw Error(PT_ERR_UNSUPPORTED, "Scene::traverse: more than 0xF0000000 rays in one batch");
        dev_->check(pt_trace_closest(dev_->handle(), scene_, rays, start, (uint32_t)n, hits));
    }
    // Batches already in HBM, without waiting for each: the batches of consecutive samples (main.rs:64-76) are independent, so up
    // to PT_TRACE_SLOTS of them may be in flight; submit on alternating slots, wait on a slot before reusing its buffers.
    void traverse_device_submit(const Ray* d_rays, const uint32_t* d_start, size_t n, Hit* d_hits, uint32_t slot) const {
        if (n > 0xF0000000ull) throw Error(PT_ERR_UNSUPPORTED, "Scene::traverse_device_submit: more than 0xF0000000 rays in one batch");
        dev_->check(pt_trace_closest_dev_submit(dev_->handle(), scene_, d_rays, d_start, (uint32_t)n, d_hits, slot));
    }
    void traverse_device_wait(uint32_t slot) const { dev_->check(pt_trace_closest_dev_wait(dev_->handle(), slot)); }
    // the render closure (main.rs:58-116): X, Y, Z, 1.0 per pixel, index 4*(x + y*W) + c
    std::vector<float> render(const Camera& cam, uint32_t samples, uint64_t seed) const {
        return render_range(cam, 0, samples, samples, seed, PT_RENDER_FINAL);
    }
    std::vector<float> render(uint64_t seed) const { return render(camera, camera.samples, seed); }
    // extension: directions normalised before the sky is evaluated (PT_RENDER_NORMALIZE_SKY_DIRECTION, ptcore.h)
    std::vector<float> render_with_normalised_sky(const Camera& cam, uint32_t samples, uint64_t seed) const {
        return render_range(cam, 0, samples, samples, seed, PT_RENDER_FINAL | PT_RENDER_NORMALIZE_SKY_DIRECTION);
    }
    // samples [begin, begin+count) of `total`; PT_RENDER_SUM gives raw sums (w = count) for sharded callers
    std::vector<float> render_range(const Camera& cam, uint32_t begin, uint32_t count, uint32_t total, uint64_t seed,
                                    uint32_t mode = PT_RENDER_SUM) const {
        std::vector<float> xyzw((size_t)cam.resolution[0] * cam.resolution[1] * 4);
        const pt_camera c = cam.ffi();
        dev_->check(pt_render(dev_->handle(), scene_, &c, begin, count, total, seed, mode, xyzw.data()));
        return xyzw;
    }
    // the same into a DEVICE buffer of W*H*4 floats (pt_render_dev): what a rank hands to ncclAllReduce without a trip
    // through host memory (harness/ptrender_mgpu.cpp)
    void render_range_dev(const Camera& cam, uint32_t begin, uint32_t count, uint32_t total, uint64_t seed, uint32_t mode,
                          float* d_xyzw) const {
        const pt_camera c = cam.ffi();
        dev_->check(pt_render_dev(dev_->handle(), scene_, &c, begin, count, total, seed, mode, d_xyzw));
    }

private:
    explicit Scene(Device& dev) : dev_(&dev) {}
    Device* dev_;
    pt_scene* scene_ = nullptr;
};

// ---- sample-split multi-GPU rendering (one process per GPU) -------------------------------------
// Rank g renders sample_range(total, g, world) with Scene::render_range (PT_RENDER_SUM: raw XYZ sums, w = samples in
// the range), the W*H*4 float buffers are summed with one collective (ncclAllReduce / MPI_Allreduce, ncclFloat, sum),
// and finalize_sums turns the total into the reference's image (main.rs:49,100).  Sample s of pixel p draws from
// stream (seed, p, s) whatever the split, so the result does not depend on the number of ranks beyond f32 summation order.
inline std::pair<uint32_t, uint32_t> sample_range(uint32_t spp_total, uint32_t rank, uint32_t world_size) {
    if (world_size == 0 || rank >= world_size) throw Error(PT_ERR_INVALID, "sample_range: bad rank / world size");
    const uint32_t base = spp_total / world_size, rem = spp_total % world_size;
    return {rank * base + (rank < rem ? rank : rem), base + (rank < rem ? 1u : 0u)};       // {begin, count}
}
inline void finalize_sums(std::vector<float>& xyzw) {
    for (size_t p = 0; p + 3 < xyzw.size(); p += 4) {
        const float inv = 1.0f / xyzw[p + 3];
        xyzw[p] *= inv; xyzw[p + 1] *= inv; xyzw[p + 2] *= inv; xyzw[p + 3] = 1.0f;
    }
}

struct Display {
    std::vector<uint8_t> bgra8;        // what the Bgra8Unorm surface holds (renderer.rs:191)
    float avg_luminance = 0, max_luminance = 0, middle_gray = 0;
};
// the window's display transform (renderer.rs:297-309, shader.wgsl:32-44), defaults renderer.rs:221-222
inline Display tonemap(Device& dev, const std::vector<float>& xyzw, uint32_t width, uint32_t height,
                       float avg_zone = 5.0f, float white_zone = 10.0f, uint32_t flags = 0) {
    if (xyzw.size() != (size_t)width * height * 4) throw Error(PT_ERR_INVALID, "tonemap: buffer size does not match the resolution");
    Display d;
    d.bgra8.resize((size_t)width * height * 4);
    float st[3] = {0, 0, 0};
    dev.check(pt_tonemap(dev.handle(), xyzw.data(), width, height, avg_zone, white_zone, flags, nullptr, d.bgra8.data(), st));
    d.avg_luminance = st[0]; d.max_luminance = st[1]; d.middle_gray = st[2];
    return d;
}

}  // namespace pathtracer

#endif  // PTCORE_HPP
