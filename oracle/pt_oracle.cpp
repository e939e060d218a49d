// pt_oracle.cpp — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the checker the HIP path is compared against.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it; the product (pathtracer_amd/) never does.
//
// It follows andreasots/pathtracer line by line (citations are file:line under /root/reference):
//   src/triangle.rs:42-63,82-122      Triangle::new, intersect, aabb
//   src/bvh.rs:28,42-109,123-448      AABB, BVHNode::build, traverse_recursive, Ray, Bucket
//   src/camera.rs:11-39               Camera
//   src/color.rs:41-58,72-93          reflectance_at, from_wavelength
//   src/material.rs:86-100,129-189,255-309   Texture::sample, D65, Lambert, Emit, Material::sample
//   src/hosek_wilkie.rs:36-384        HosekWilkieSkyModel (tables: pathtracer_amd/data/*.bin)
//   src/scene.rs:105-129,284-374      InitialisedSky::from_sky, evaluate_sky, radiance
//   src/main.rs:22-26,64-109          sample loop
//
// The reference cannot be built here (Rust; no toolchain, no vendored crates), so there is no
// oracle/_ref.  PINNING: the only known-answer vectors the reference holds for this path are the
// five Triangle KATs (triangle.rs:130-216), the camera corner directions up to scale
// (camera.rs:47-66) and the sRGB->XYZ primaries (color.rs:184-218); tests/test_oracle_kat.py
// checks all of them.  BVH build, traversal order, materials, spectra, sky and the integrator
// have no reference vectors: for those this oracle is "parity unpinned"  and rests
// on being a line-faithful restatement.
//
// Deliberate differences, by design of the build (north_star):
//   * RNG: rng.rs's entropy-seeded thread-local xoshiro256++ is replaced by the counter-based
//     per-(pixel, sample) stream of include/pt_rng.h; the distributions and the draw order are
//     the reference's (StandardNormal's ziggurat is replaced by an inverse-CDF sampler).
//   * exp/ln/acos/sin/x^1.5/cbrt on the per-ray path come from include/pt_math.h so that CPU and
//     GPU agree bit for bit; build with -DPTO_USE_LIBM to use the platform libm instead (used by
//     tests to bound the difference).
// Arithmetic order follows nalgebra 0.31.3 (dot = (x*x' + y*y') + z*z', lerp = a*(1-t) + b*t,
// normalize = component-wise division by the norm); compile with -ffp-contract=off.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "../include/ptcore.h"
#include "../include/pt_rng.h"

extern "C" {
extern const float pto_tab_hw_datasets[];      // [11][2][10][6][9]   hosek_wilkie.rs:387
extern const float pto_tab_hw_datasets_rad[];  // [11][2][10][6]      hosek_wilkie.rs:2216
extern const float pto_tab_hw_solar[];         // [11][10][45][4]     hosek_wilkie.rs:2505
extern const float pto_tab_hw_limb[];          // [11][6]             hosek_wilkie.rs:7700
extern const float pto_tab_srgb2refl[];        // [471][3]            build.rs:36-41
extern const float pto_tab_d65[];              // [531]               build.rs:43-50
}

namespace {

#ifdef PTO_USE_LIBM
inline float m_exp(float x) { return std::exp(x); }
inline float m_acos(float x) { return std::acos(x); }
inline float m_sin(float x) { return std::sin(x); }
inline float m_pow1p5(float x) { return std::pow(x, 1.5f); }
inline float m_cbrt(float x) { return std::cbrt(x); }
#else
inline float m_exp(float x) { return pt_expf(x); }
inline float m_acos(float x) { return pt_acosf(x); }
inline float m_sin(float x) { return pt_sinf(x); }
inline float m_pow1p5(float x) { return pt_pow1p5(x); }
inline float m_cbrt(float x) { return pt_cbrtf(x); }
#endif

constexpr float INF = std::numeric_limits<float>::infinity();

// ---- nalgebra Vector3 semantics ------------------------------------------------------------
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float norm_squared(V3 a) { return dot(a, a); }
inline V3 normalize(V3 a) { return a / std::sqrt(norm_squared(a)); }
inline V3 v3(const float* p) { return {p[0], p[1], p[2]}; }

inline float fmin_rs(float a, float b) { return pt_fmin(a, b); }   // f32::min
inline float fmax_rs(float a, float b) { return pt_fmax(a, b); }   // f32::max

// ---- bvh.rs:32-109 AABB ---------------------------------------------------------------------
struct AABB {
    V3 min, max;
    static AABB empty() { return {{INF, INF, INF}, {-INF, -INF, -INF}}; }               // :47-52
    AABB join(const AABB& o) const {                                                      // :54-67
        return {{fmin_rs(min.x, o.min.x), fmin_rs(min.y, o.min.y), fmin_rs(min.z, o.min.z)},
                {fmax_rs(max.x, o.max.x), fmax_rs(max.y, o.max.y), fmax_rs(max.z, o.max.z)}};
    }
    AABB grow(V3 p) const {                                                               // :82-87
        return {{fmin_rs(min.x, p.x), fmin_rs(min.y, p.y), fmin_rs(min.z, p.z)},
                {fmax_rs(max.x, p.x), fmax_rs(max.y, p.y), fmax_rs(max.z, p.z)}};
    }
    V3 size() const { return max - min; }                                                 // :89-91
    V3 center() const { return min + (size() / 2.0f); }                                   // :93-95
    bool is_empty() const { return min.x > max.x || min.y > max.y || min.z > max.z; }     // :97-99
    float surface_area() const {                                                          // :101-104
        V3 s = size();
        return 2.0f * (s.x * s.y + s.x * s.z + s.y * s.z);
    }
    int largest_axis() const {                                                            // :106-108
        V3 s = size();   // nalgebra argmax: first maximum, strict '>' scan from index 0
        int i = 0; float m = s.x;
        if (s.y > m) { m = s.y; i = 1; }
        if (s.z > m) { m = s.z; i = 2; }
        return i;
    }
};

// ---- bvh.rs:372-411 Ray ----------------------------------------------------------------------
struct Ray {
    V3 origin, direction, inv_direction;
    Ray() = default;
    Ray(V3 o, V3 d) : origin(o), direction(d), inv_direction{1.0f / d.x, 1.0f / d.y, 1.0f / d.z} {}
    // :390-410; returns false for None, *tmin_out = Some(tmin)
    bool intersects_aabb(const AABB& b, float* tmin_out) const {
        V3 t1 = cmul(b.min - origin, inv_direction);
        V3 t2 = cmul(b.max - origin, inv_direction);
        float tmin = fmin_rs(t1.x, t2.x);
        float tmax = fmax_rs(t1.x, t2.x);
        tmin = fmax_rs(tmin, fmin_rs(t1.y, t2.y));
        tmax = fmin_rs(tmax, fmax_rs(t1.y, t2.y));
        tmin = fmax_rs(tmin, fmin_rs(t1.z, t2.z));
        tmax = fmin_rs(tmax, fmax_rs(t1.z, t2.z));
        tmin = fmax_rs(tmin, 0.0f);
        if (tmax >= tmin) { *tmin_out = tmin; return true; }
        return false;
    }
};

// ---- triangle.rs:6-33 ------------------------------------------------------------------------
struct Triangle {
    V3 a, b, c;
    bool has_uv, has_normals;
    float uv[3][2];
    V3 vn[3];
    uint32_t material_index;
    V3 n; float d;
    V3 n1; float d1;
    V3 n2; float d2;
};
struct Intersection { float distance, u, v; V3 normal; };

// triangle.rs:42-63
Triangle triangle_new(const pt_triangle& t) {
    Triangle r{};
    r.a = v3(t.a); r.b = v3(t.b); r.c = v3(t.c);
    r.has_uv = (t.flags & PT_TRI_HAS_UV) != 0;
    r.has_normals = (t.flags & PT_TRI_HAS_NORMALS) != 0;
    for (int i = 0; i < 3; ++i) {
        r.uv[i][0] = t.uv[i][0]; r.uv[i][1] = t.uv[i][1];
        r.vn[i] = v3(t.vertex_normals[i]);
    }
    r.material_index = t.material_index;
    V3 ab = r.b - r.a;
    V3 ac = r.c - r.a;
    r.n = cross(ab, ac);
    r.d = dot(r.a, r.n);
    r.n1 = cross(ac, r.n) / norm_squared(r.n);
    r.d1 = -dot(r.a, r.n1);
    r.n2 = cross(r.n, ab) / norm_squared(r.n);
    r.d2 = -dot(r.a, r.n2);
    return r;
}

// triangle.rs:82-115
// Set when a triangle test took the SIGN of a NaN (triangle.rs:91,97,102: f32::is_sign_positive reads the sign bit).  The
// sign of a NaN that an operation generates or passes on is not specified by IEEE 754 and not portable in Rust either
// (std docs of is_sign_positive); x86 SSE and gfx950 differ, so on such rays the reference's own answer depends on the CPU
// it runs on.  pto_trace_nan_flags reports them so that parity tests can leave them out, and say how many there were.
thread_local bool tl_nan_sign_test = false;
inline bool sign_differs(float a, float b) {
    if (a != a || b != b) tl_nan_sign_test = true;
    return pt_sign_positive(a) != pt_sign_positive(b);
}

bool triangle_intersect(const Triangle& tr, const Ray& ray, float max_distance, Intersection* out) {
    float det = dot(ray.direction, tr.n);
    if (det == 0.0f) return false;
    float t = tr.d - dot(ray.origin, tr.n);
    if (sign_differs(max_distance * det - t, t)) return false;
    V3 p = ray.origin * det + ray.direction * t;
    float u = dot(p, tr.n1) + det * tr.d1;
    if (sign_differs(det - u, u)) return false;
    float v = dot(p, tr.n2) + det * tr.d2;
    if (sign_differs(det - u - v, v)) return false;
    float inv_det = 1.0f / det;
    float distance = t * inv_det;
    // approx 0.5.1 relative_eq!(distance, 0.0) with default epsilon = max_relative = f32::EPSILON:
    // equal, or |diff| <= epsilon; the relative branch |x| <= |x|*eps never holds for x != 0.
    if (distance == 0.0f || std::fabs(distance) <= 1.1920928955078125e-7f) return false;
    out->distance = distance;
    out->normal = tr.n;
    out->u = u * inv_det;
    out->v = v * inv_det;
    return true;
}

// triangle.rs:118-122
AABB triangle_aabb(const Triangle& t) { return AABB{t.a, t.a}.grow(t.b).grow(t.c); }

// ---- bvh.rs:123-248 build --------------------------------------------------------------------
struct Node {
    bool is_leaf; uint32_t shape_index;
    uint32_t child_l_index, child_r_index;
    AABB child_l_aabb, child_r_aabb;
};

constexpr float EPSILON_BVH = 0.00001f;   // bvh.rs:28

struct Bucket {                            // bvh.rs:420-439
    size_t size; AABB aabb;
    static Bucket empty() { return {0, AABB::empty()}; }
    void add_aabb(const AABB& a) { size += 1; aabb = aabb.join(a); }
    static Bucket join_bucket(Bucket a, const Bucket& b) { return {a.size + b.size, a.aabb.join(b.aabb)}; }
};

AABB joint_aabb_of_shapes(const std::vector<uint32_t>& idx, size_t lo, size_t hi,
                          const std::vector<Triangle>& shapes) {                          // :441-448
    AABB a = AABB::empty();
    for (size_t i = lo; i < hi; ++i) a = a.join(triangle_aabb(shapes[idx[i]]));
    return a;
}

struct BuildError { };

uint32_t build_rec(const std::vector<Triangle>& shapes, const std::vector<uint32_t>& indices,
                   std::vector<Node>& nodes) {                                            // :134-248
    AABB aabb_bounds = AABB::empty(), centroid_bounds = AABB::empty();
    for (uint32_t index : indices) {                                                      // :147-151
        AABB sa = triangle_aabb(shapes[index]);
        V3 center = sa.center();
        aabb_bounds = aabb_bounds.join(sa);
        centroid_bounds = centroid_bounds.grow(center);
    }
    if (indices.size() == 1) {                                                            // :154-159
        uint32_t node_index = (uint32_t)nodes.size();
        nodes.push_back(Node{true, indices[0], 0, 0, AABB::empty(), AABB::empty()});
        return node_index;
    }
    uint32_t node_index = (uint32_t)nodes.size();                                         // :163-164
    nodes.push_back(Node{true, 0, 0, 0, AABB::empty(), AABB::empty()});

    int split_axis = centroid_bounds.largest_axis();                                      // :167
    float split_axis_size = centroid_bounds.max[split_axis] - centroid_bounds.min[split_axis];

    uint32_t child_l_index, child_r_index;
    AABB child_l_aabb, child_r_aabb;
    if (split_axis_size < EPSILON_BVH) {                                                  // :171-183
        size_t mid = indices.size() / 2;
        std::vector<uint32_t> l(indices.begin(), indices.begin() + mid);
        std::vector<uint32_t> r(indices.begin() + mid, indices.end());
        child_l_aabb = joint_aabb_of_shapes(l, 0, l.size(), shapes);
        child_r_aabb = joint_aabb_of_shapes(r, 0, r.size(), shapes);
        child_l_index = build_rec(shapes, l, nodes);
        child_r_index = build_rec(shapes, r, nodes);
    } else {
        constexpr int NUM_BUCKETS = 6;                                                    // :186
        Bucket buckets[NUM_BUCKETS];
        for (auto& b : buckets) b = Bucket::empty();
        std::vector<uint32_t> assign[NUM_BUCKETS];
        for (uint32_t idx : indices) {                                                    // :192-207
            AABB sa = triangle_aabb(shapes[idx]);
            V3 sc = sa.center();
            float rel = (sc[split_axis] - centroid_bounds.min[split_axis]) / split_axis_size;
            uint32_t bn = pt_f2usize(rel * ((float)NUM_BUCKETS - 0.01f));
            if (bn >= (uint32_t)NUM_BUCKETS) throw BuildError{};   // Rust: index out of bounds panic
            buckets[bn].add_aabb(sa);
            assign[bn].push_back(idx);
        }
        int min_bucket = 0;                                                               // :210-228
        float min_cost = INF;
        child_l_aabb = AABB::empty();
        child_r_aabb = AABB::empty();
        for (int i = 0; i < NUM_BUCKETS - 1; ++i) {
            Bucket cl = Bucket::empty(), cr = Bucket::empty();
            for (int k = 0; k <= i; ++k) cl = Bucket::join_bucket(cl, buckets[k]);
            for (int k = i + 1; k < NUM_BUCKETS; ++k) cr = Bucket::join_bucket(cr, buckets[k]);
            float cost = ((float)cl.size * cl.aabb.surface_area() + (float)cr.size * cr.aabb.surface_area())
                         / aabb_bounds.surface_area();
            if (cost < min_cost) {
                min_bucket = i; min_cost = cost;
                child_l_aabb = cl.aabb; child_r_aabb = cr.aabb;
            }
        }
        std::vector<uint32_t> l, r;                                                       // :231-233
        for (int k = 0; k <= min_bucket; ++k) l.insert(l.end(), assign[k].begin(), assign[k].end());
        for (int k = min_bucket + 1; k < NUM_BUCKETS; ++k) r.insert(r.end(), assign[k].begin(), assign[k].end());
        for (auto& a : assign) { a.clear(); a.shrink_to_fit(); }
        // An empty side would recurse forever in the reference (indices.len()==0 is never a leaf);
        // it cannot happen when a split was chosen, and is caught by the asserts below otherwise.
        if (child_l_aabb.is_empty() || child_r_aabb.is_empty()) throw BuildError{};       // :242-243
        child_l_index = build_rec(shapes, l, nodes);                                      // :236-237
        child_r_index = build_rec(shapes, r, nodes);
    }
    if (child_l_aabb.is_empty() || child_r_aabb.is_empty()) throw BuildError{};           // :242-243
    nodes[node_index] = Node{false, 0, child_l_index, child_r_index, child_l_aabb, child_r_aabb};
    return node_index;
}

// ---- bvh.rs:250-338 traverse_recursive, with visit counters -----------------------------------
struct Counters { uint64_t nodes = 0, tris = 0; };

struct TravResult { bool some; uint32_t tri; Intersection isect; };

TravResult traverse_rec(const std::vector<Node>& nodes, uint32_t node_index, const Ray& ray,
                        uint32_t start, const std::vector<Triangle>& shapes, float max_distance,
                        Counters& cnt) {
    const Node& node = nodes[node_index];
    if (!node.is_leaf) {
        cnt.nodes += 1;
        float l_min = 0, r_min = 0;
        bool left = ray.intersects_aabb(node.child_l_aabb, &l_min);                       // :260
        bool right = ray.intersects_aabb(node.child_r_aabb, &r_min);                      // :261
        if (left && right) {                                                              // :264
            uint32_t first, second; float second_min;
            if (l_min < r_min) { first = node.child_l_index; second = node.child_r_index; second_min = r_min; }
            else               { first = node.child_r_index; second = node.child_l_index; second_min = l_min; }
            TravResult f = traverse_rec(nodes, first, ray, start, shapes, max_distance, cnt);
            float fd = f.some ? f.isect.distance : INF;                                   // :280-283
            if (fd > second_min) {                                                        // :285
                TravResult s = traverse_rec(nodes, second, ray, start, shapes,
                                            fmin_rs(fd, max_distance), cnt);
                float sd = s.some ? s.isect.distance : INF;
                return (fd < sd) ? f : s;                                                 // :299-303
            }
            return f;                                                                     // :305
        } else if (left) {
            return traverse_rec(nodes, node.child_l_index, ray, start, shapes, max_distance, cnt);
        } else if (right) {
            return traverse_rec(nodes, node.child_r_index, ray, start, shapes, max_distance, cnt);
        }
        return TravResult{false, PT_NO_TRIANGLE, {}};                                     // :324
    }
    // Leaf :327-336 — `start` is compared by identity; max_distance is NOT used (:331-332)
    if (node.shape_index != start) {
        cnt.tris += 1;
        Intersection is;
        if (triangle_intersect(shapes[node.shape_index], ray, INF, &is))
            return TravResult{true, node.shape_index, is};
    }
    return TravResult{false, PT_NO_TRIANGLE, {}};
}

// ---- camera.rs:11-29 --------------------------------------------------------------------------
struct Camera {
    float csx, csy, s0, s1, focal;
    Camera(float w, float h, float sensor_width, float focal_length)
        : csx(1.0f / (w + 1.0f)), csy(1.0f / (h + 1.0f)), s0(sensor_width),
          s1(sensor_width * h / w), focal(focal_length) {}
    Ray generate_ray(float x, float y) const {
        return Ray({0, 0, 0}, {((x + 0.5f) * csx - 0.5f) * s0, -((y + 0.5f) * csy - 0.5f) * s1, -focal});
    }
};

// ---- color.rs ---------------------------------------------------------------------------------
constexpr int NW = 8;                      // COLOR_SAMPLES main.rs:26
constexpr float MIN_WAVELENGTH = 360.0f;   // main.rs:22
constexpr float MAX_WAVELENGTH = 830.0f;   // main.rs:23

inline V3 refl_row(uint32_t i) {           // TABLE.get(i).copied().unwrap_or([0.0; 3])
    if (i < 471u) return {pto_tab_srgb2refl[3 * i], pto_tab_srgb2refl[3 * i + 1], pto_tab_srgb2refl[3 * i + 2]};
    return {0, 0, 0};
}
float reflectance_at_one(V3 rgb, float wavelength) {                                      // color.rs:41-54
    float offset = wavelength - 360.0f;
    float index = std::floor(offset);
    float alpha = offset - index;
    uint32_t i = pt_f2usize(index);
    V3 b0 = refl_row(i);
    V3 b1 = (i == 0x80000000u) ? V3{0, 0, 0} : refl_row(i + 1);
    V3 base = b0 * (1.0f - alpha) + b1 * alpha;    // nalgebra lerp
    return dot(rgb, base);
}
void from_wavelength(float wl, float out[3]) {                                            // color.rs:72-93
    auto gauss = [](float w, float weight, float mean, float s1, float s2) {
        float t = (w - mean) * (w < mean ? s1 : s2);
        return weight * m_exp(-0.5f * (t * t));
    };
    out[0] = gauss(wl, 0.362f, 442.0f, 0.0624f, 0.0374f) + gauss(wl, 1.056f, 599.8f, 0.0264f, 0.0323f)
             + gauss(wl, -0.065f, 501.1f, 0.0490f, 0.0382f);
    out[1] = gauss(wl, 0.821f, 568.8f, 0.0213f, 0.0247f) + gauss(wl, 0.286f, 530.9f, 0.0613f, 0.0322f);
    out[2] = gauss(wl, 1.217f, 437.0f, 0.0845f, 0.0278f) + gauss(wl, 0.681f, 459.0f, 0.0385f, 0.0725f);
}
// color.rs:108-117 (host-side texture luminance only; kept for the reference's colour KATs)
void srgb_to_xyz(const float rgb[3], float out[3]) {
    const float m[3][3] = {{0.41239080f, 0.35758434f, 0.18048079f},
                           {0.21263901f, 0.71516868f, 0.07219232f},
                           {0.01933082f, 0.11919478f, 0.95053215f}};
    for (int i = 0; i < 3; ++i) out[i] = (m[i][0] * rgb[0] + m[i][1] * rgb[1]) + m[i][2] * rgb[2];
}

// ---- material.rs ------------------------------------------------------------------------------
float d65_sample_one(float wavelength) {                                                  // :132-141
    float offset = std::floor(wavelength) - 300.0f;
    float index = std::floor(offset);
    float alpha = offset - index;
    uint32_t i = pt_f2usize(index);
    float t0 = (i < 531u) ? pto_tab_d65[i] : 0.0f;
    float t1 = (i != 0x80000000u && i + 1 < 531u) ? pto_tab_d65[i + 1] : 0.0f;
    return t0 * (1.0f - alpha) + t1 * alpha;
}

inline float rem_euclid1(float x) {        // f32::rem_euclid(1.0)
    float r = std::fmod(x, 1.0f);
    return (r < 0.0f) ? r + 1.0f : r;
}

struct Scene;
// Texture::sample :90-100 (Nearest) / :101-123 (Bilinear) / :124 (Flat).  channels==3 -> colour, 1 -> value in .x
V3 texture_sample(const Scene& sc, const pt_texture& t, float u, float v);

struct Sample { float emit[NW]; float weight[NW]; Ray new_ray; };

// ---- hosek_wilkie.rs --------------------------------------------------------------------------
inline float powi(float x, int n) {        // llvm.powi / compiler-rt __powisf2: square-and-multiply
    float r = 1.0f; bool first = true;
    for (;;) {
        if (n & 1) { r = first ? x : r * x; first = false; }
        n /= 2;
        if (n == 0) break;
        x = x * x;
    }
    return r;
}

inline float ds(int wl, int alb, int turb, int k, int i) {   // DATASETS[wl][alb][turb][k][i]
    return pto_tab_hw_datasets[(((wl * 2 + alb) * 10 + turb) * 6 + k) * 9 + i];
}
inline float dsr(int wl, int alb, int turb, int k) {         // DATASETS_RAD[wl][alb][turb][k]
    return pto_tab_hw_datasets_rad[((wl * 2 + alb) * 10 + turb) * 6 + k];
}

// the quintic Bernstein blend written out at hosek_wilkie.rs:160-171 (and 7 more times)
template <class F> inline float bern5(float se, F e) {
    return powi(1.0f - se, 5) * e(0) + 5.0f * powi(1.0f - se, 4) * se * e(1)
           + 10.0f * powi(1.0f - se, 3) * powi(se, 2) * e(2)
           + 10.0f * powi(1.0f - se, 2) * powi(se, 3) * e(3)
           + 5.0f * (1.0f - se) * powi(se, 4) * e(4) + powi(se, 5) * e(5);
}

struct HosekWilkie { float configs[11][9]; float radiances[11]; float turbidity; float solar_radius; };

void hw_cook_configuration(int wl, float turbidity, float albedo, float solar_elevation, float config[9]) {  // :144-229
    int int_turbidity = (int)pt_f2usize(turbidity);
    float turbidity_rem = turbidity - (float)int_turbidity;
    float se = std::cbrt(solar_elevation / 1.57079632679489661923f);
    for (int i = 0; i < 9; ++i)
        config[i] = (1.0f - albedo) * (1.0f - turbidity_rem)
                    * bern5(se, [&](int k) { return ds(wl, 0, int_turbidity - 1, k, i); });
    for (int i = 0; i < 9; ++i)
        config[i] += albedo * (1.0f - turbidity_rem)
                     * bern5(se, [&](int k) { return ds(wl, 1, int_turbidity - 1, k, i); });
    if (int_turbidity == 10) return;
    for (int i = 0; i < 9; ++i)
        config[i] += (1.0f - albedo) * turbidity_rem
                     * bern5(se, [&](int k) { return ds(wl, 0, int_turbidity, k, i); });
    for (int i = 0; i < 9; ++i)
        config[i] += albedo * turbidity_rem
                     * bern5(se, [&](int k) { return ds(wl, 1, int_turbidity, k, i); });
}
float hw_cook_radiance_configuration(int wl, float turbidity, float albedo, float solar_elevation) {  // :231-304
    int int_turbidity = (int)pt_f2usize(turbidity);
    float turbidity_rem = turbidity - (float)int_turbidity;
    float se = std::cbrt(solar_elevation / 1.57079632679489661923f);
    float res = (1.0f - albedo) * (1.0f - turbidity_rem)
                * bern5(se, [&](int k) { return dsr(wl, 0, int_turbidity - 1, k); });
    res += albedo * (1.0f - turbidity_rem) * bern5(se, [&](int k) { return dsr(wl, 1, int_turbidity - 1, k); });
    if (int_turbidity == 10) return res;
    res += (1.0f - albedo) * turbidity_rem * bern5(se, [&](int k) { return dsr(wl, 0, int_turbidity, k); });
    res += albedo * turbidity_rem * bern5(se, [&](int k) { return dsr(wl, 1, int_turbidity, k); });
    return res;
}
HosekWilkie hw_new(float solar_elevation, float turbidity, float albedo) {                // :46-72
    HosekWilkie s{};
    s.turbidity = turbidity;
    s.solar_radius = (0.51f * (3.14159265358979323846f / 180.0f)) / 2.0f;   // 0.51f32.to_radians() / 2.0
    for (int wl = 0; wl < 11; ++wl) {
        hw_cook_configuration(wl, turbidity, albedo, solar_elevation, s.configs[wl]);
        s.radiances[wl] = hw_cook_radiance_configuration(wl, turbidity, albedo, solar_elevation);
    }
    return s;
}
float hw_get_radiance_internal(const float c[9], float cos_theta, float gamma, float cos_gamma) {  // :306-325
    float exp_m = m_exp(c[4] * gamma);
    float ray_m = cos_gamma * cos_gamma;
    float mie_m = (1.0f + cos_gamma * cos_gamma) / m_pow1p5(1.0f + c[8] * c[8] - 2.0f * c[8] * cos_gamma);
    float zenith = std::sqrt(cos_theta);
    return (1.0f + c[0] * m_exp(c[1] / (cos_theta + 0.01f)))
           * (c[2] + c[3] * exp_m + c[5] * ray_m + c[6] * mie_m + c[7] * zenith);
}
float hw_radiance_one(const HosekWilkie& m, float cos_theta, float gamma, float cos_gamma, float wl) {  // :74-102
    uint32_t low_wl = pt_f2usize((wl - 320.0f) / 40.0f);
    if (low_wl >= 11u) return 0.0f;
    float interp = pt_fract_rem((wl - 320.0f) / 40.0f);
    float val_low = hw_get_radiance_internal(m.configs[low_wl], cos_theta, gamma, cos_gamma) * m.radiances[low_wl] * 1.0f;
    if (interp < 1e-6f) return val_low;
    float result = (1.0f - interp) * val_low;
    if (low_wl + 1 < 11u)
        result += interp * hw_get_radiance_internal(m.configs[low_wl + 1], cos_theta, gamma, cos_gamma)
                  * m.radiances[low_wl + 1] * 1.0f;
    return result;
}
float hw_sr_internal(uint32_t turbidity, uint32_t wl, float elevation) {                  // :327-345
    constexpr uint32_t PIECES = 45;
    uint32_t pos = std::min(pt_f2usize(m_cbrt(2.0f * elevation / 3.14159265358979323846f) * (float)PIECES), PIECES - 1);
    float break_x = powi((float)pos / (float)PIECES, 3) * 1.57079632679489661923f;
    float res = 0.0f;
    float x = elevation - break_x;
    float x_exp = 1.0f;
    const float* coefs = &pto_tab_hw_solar[((wl * 10 + turbidity) * 45 + pos) * 4];
    for (int k = 3; k >= 0; --k) { res += x_exp * coefs[k]; x_exp *= x; }
    return res * 1.0f;
}
float hw_solar_radiance_internal2(const HosekWilkie& m, float wl, float elevation, float gamma) {  // :347-383
    uint32_t turb_low = std::min(pt_f2usize(m.turbidity) - 1u, 8u);
    float turb_frac = m.turbidity - (float)(turb_low + 1);
    uint32_t wl_low = std::min(pt_f2usize((wl - 320.0f) / 40.0f), 9u);
    float wl_frac = (wl - 320.0f - (float)wl_low * 40.0f) / 40.0f;
    float direct_radiance =
        (1.0f - turb_frac) * ((1.0f - wl_frac) * hw_sr_internal(turb_low, wl_low, elevation)
                              + wl_frac * hw_sr_internal(turb_low, wl_low + 1, elevation))
        + turb_frac * ((1.0f - wl_frac) * hw_sr_internal(turb_low + 1, wl_low, elevation)
                       + wl_frac * hw_sr_internal(turb_low + 1, wl_low + 1, elevation));
    float ld[6];
    for (int i = 0; i < 6; ++i)
        ld[i] = (1.0f - wl_frac) * pto_tab_hw_limb[wl_low * 6 + i] + wl_frac * pto_tab_hw_limb[(wl_low + 1) * 6 + i];
    float sol_rad_sin = m_sin(m.solar_radius);
    float ar2 = 1.0f / (sol_rad_sin * sol_rad_sin);
    float singamma = m_sin(gamma);
    float sc = std::sqrt(fmax_rs(1.0f - ar2 * singamma * singamma, 0.0f));
    float darkening = ld[0] + ld[1] * sc + ld[2] * powi(sc, 2) + ld[3] * powi(sc, 3) + ld[4] * powi(sc, 4)
                      + ld[5] * powi(sc, 5);
    return direct_radiance * darkening;
}
float hw_solar_radiance_one(const HosekWilkie& m, float theta, float cos_theta, float gamma, float cos_gamma, float wl) {  // :114-131
    if (wl < 320.0f || wl > 720.0f) return 0.0f;
    float direct = hw_solar_radiance_internal2(m, wl, 1.57079632679489661923f - theta, gamma);
    float inscattered = hw_radiance_one(m, cos_theta, gamma, cos_gamma, wl);
    return direct + inscattered;
}

// ---- scene ------------------------------------------------------------------------------------
struct Image { uint32_t w, h, ch; std::vector<float> texels; };

struct Scene {
    std::vector<Triangle> triangles;
    std::vector<Node> nodes;
    std::vector<pt_material> materials;
    std::vector<Image> images;
    pt_sky sky;
    HosekWilkie hw;
};

V3 texture_sample(const Scene& sc, const pt_texture& t, float u, float v) {
    if (t.kind == PT_TEX_FLAT) return {t.value[0], t.value[1], t.value[2]};               // :124
    const Image& im = sc.images[t.image];
    auto texel = [&](uint32_t x, uint32_t y) -> V3 {                                      // Index<(u32, u32)> :28-37
        const float* p = &im.texels[((size_t)(x % im.w) + (size_t)(y % im.h) * im.w) * im.ch];
        if (im.ch == 3) return {p[0], p[1], p[2]};
        return {p[0], p[0], p[0]};
    };
    if (t.kind == PT_TEX_NEAREST) {                                                       // :92-100
        uint32_t x = pt_f2usize(std::floor(rem_euclid1(u) * (float)im.w));
        uint32_t y = pt_f2usize(std::floor((1.0f - rem_euclid1(v)) * (float)im.h));
        return texel(x, y);
    }
    // Texture::Bilinear :101-123
    const float x = rem_euclid1(u) * (float)im.w;                                         // :105
    const float y = (1.0f - rem_euclid1(v)) * (float)im.h;                                // :106
    const uint32_t x0 = pt_f2usize(std::floor(x)), y0 = pt_f2usize(std::floor(y));        // :108-109
    const float t_x = x - std::floor(x), t_y = y - std::floor(y);                         // :111-112
    const V3 p00 = texel(x0, y0), p10 = texel(x0 + 1u, y0), p01 = texel(x0, y0 + 1u), p11 = texel(x0 + 1u, y0 + 1u);   // :114-117
    V3 out;
    const float* a = &p00.x; const float* b = &p10.x; const float* c = &p01.x; const float* d = &p11.x; float* o = &out.x;
    for (int k = 0; k < 3; ++k) {
        const float p0 = a[k] * (1.0f - t_x) + b[k] * t_x;                                // :119
        const float p1 = c[k] * (1.0f - t_x) + d[k] * t_x;                                // :120
        o[k] = p0 * (1.0f - t_y) + p1 * t_y;                                              // :122
    }
    return out;
}

// scene.rs:284-305
void evaluate_sky(const Scene& sc, const Ray& ray, const float wl[NW], float out[NW]) {
    if (sc.sky.kind == PT_SKY_D65) {
        if (sc.sky.power == 0.0f) { for (int i = 0; i < NW; ++i) out[i] = 0.0f; return; }
        for (int i = 0; i < NW; ++i) out[i] = d65_sample_one(wl[i]) * sc.sky.power;
        return;
    }
    float cos_theta = dot(ray.direction, v3(sc.sky.zenith));
    if (cos_theta > 0.0f) {
        float theta = m_acos(fmin_rs(cos_theta, 1.0f));
        float cos_gamma = dot(ray.direction, v3(sc.sky.sun));
        float gamma = m_acos(fmax_rs(fmin_rs(cos_gamma, 1.0f), -1.0f));
        if (gamma < sc.hw.solar_radius) {
            for (int i = 0; i < NW; ++i) out[i] = hw_solar_radiance_one(sc.hw, theta, cos_theta, gamma, cos_gamma, wl[i]);
        } else {
            for (int i = 0; i < NW; ++i) out[i] = hw_radiance_one(sc.hw, cos_theta, gamma, cos_gamma, wl[i]);
        }
    } else {
        for (int i = 0; i < NW; ++i) out[i] = 0.0f;
    }
}

// material.rs:255-297; returns false for None (dissolved)
bool material_sample(const Scene& sc, const pt_material& mat, const Ray& ray, const float wl[NW],
                     const Intersection& is, const Triangle& tri, pt_rng* rng, Sample* out) {
    float u, v;
    if (tri.has_uv) {                                                                     // :266-273
        float w0 = 1.0f - is.u - is.v;
        u = w0 * tri.uv[0][0] + is.u * tri.uv[1][0] + is.v * tri.uv[2][0];
        v = w0 * tri.uv[0][1] + is.u * tri.uv[1][1] + is.v * tri.uv[2][1];
    } else { u = is.u; v = is.v; }
    if (pt_rng_f32(rng) >= mat.base_dissolve * texture_sample(sc, mat.dissolve, u, v).x) return false;  // :275
    V3 normal;
    if (tri.has_normals) normal = (1.0f - is.u - is.v) * tri.vn[0] + is.u * tri.vn[1] + is.v * tri.vn[2];  // :279-285
    else normal = is.normal;
    normal = (dot(normal, ray.direction) < 0.0f) ? normal : -normal;                      // :288
    normal = normalize(normal);                                                           // :290
    V3 p = ray.origin + ray.direction * is.distance;                                      // :292
    float sph[3];                                                                         // Lambert::sample :164-171
    pt_rng_unit_sphere(rng, sph);
    V3 d = normal + V3{sph[0], sph[1], sph[2]};
    V3 refl = texture_sample(sc, mat.diffuse, u, v);
    for (int i = 0; i < NW; ++i) out->weight[i] = reflectance_at_one(refl, wl[i]);
    if (mat.emit_kind == PT_EMIT_ABSORBED_D65) {                                          // :183-185
        V3 ab = texture_sample(sc, mat.absorption, u, v);
        for (int i = 0; i < NW; ++i) out->emit[i] = d65_sample_one(wl[i]) * reflectance_at_one(ab, wl[i]);
    } else {
        for (int i = 0; i < NW; ++i) out->emit[i] = 0.0f;
    }
    out->new_ray = Ray(p, d);
    return true;
}

struct PathCounters { uint64_t segments = 0; Counters trav; bool normalize_sky = false; /* PT_RENDER_NORMALIZE_SKY_DIRECTION */ };

// scene.rs:307-374
void radiance(const Scene& sc, Ray ray, const float wl[NW], pt_rng* rng, float out[NW], PathCounters& pc) {
    struct Bounce { float weight[NW]; float emit[NW]; };
    Bounce subpath[16]; int len = 0;
    uint32_t start = PT_NO_TRIANGLE;
    for (int it = 0; it < 1024; ++it) {
        if (len == 16) break;                                                             // :329-331
        TravResult tr{false, PT_NO_TRIANGLE, {}};
        pc.segments += 1;
        if (!sc.nodes.empty()) tr = traverse_rec(sc.nodes, 0, ray, start, sc.triangles, INF, pc.trav);  // bvh.rs:364-368
        if (tr.some) {
            const Triangle& tri = sc.triangles[tr.tri];
            Sample s;
            if (material_sample(sc, sc.materials[tri.material_index], ray, wl, tr.isect, tri, rng, &s)) {
                if (len > 1) {                                                            // :342-353
                    float mx = s.weight[0];      // argmax: first maximum
                    for (int i = 1; i < NW; ++i) if (s.weight[i] > mx) mx = s.weight[i];
                    if (pt_rng_f32(rng) < mx) {
                        for (int i = 0; i < NW; ++i) s.weight[i] /= mx;
                    } else {
                        for (int i = 0; i < NW; ++i) { subpath[len].weight[i] = 0.0f; subpath[len].emit[i] = s.emit[i]; }
                        len++;
                        break;
                    }
                }
                for (int i = 0; i < NW; ++i) { subpath[len].weight[i] = s.weight[i]; subpath[len].emit[i] = s.emit[i]; }
                len++;
                ray = s.new_ray;
                start = tr.tri;
            } else {                                                                      // :357-361
                ray = Ray(ray.origin + tr.isect.distance * ray.direction, ray.direction);
                start = tr.tri;
            }
        } else {                                                                          // :362-367
            float sky[NW];
            if (pc.normalize_sky) {            // extension (ptcore.h): the reference passes the direction as it is
                const float n = std::sqrt(dot(ray.direction, ray.direction));
                evaluate_sky(sc, Ray(ray.origin, {ray.direction.x / n, ray.direction.y / n, ray.direction.z / n}), wl, sky);
            } else
            evaluate_sky(sc, ray, wl, sky);
            for (int i = 0; i < NW; ++i) { subpath[len].weight[i] = 0.0f; subpath[len].emit[i] = sky[i]; }
            len++;
            break;
        }
    }
    for (int i = 0; i < NW; ++i) out[i] = 0.0f;                                           // :371-373
    for (int k = len - 1; k >= 0; --k)
        for (int i = 0; i < NW; ++i) out[i] = out[i] * subpath[k].weight[i] + subpath[k].emit[i];
}

// main.rs:76-97: one sample of one pixel, added into acc[3]
void sample_pixel(const Scene& sc, const Camera& cam, uint32_t px, uint32_t py, uint32_t width,
                  uint32_t s, uint64_t seed, float acc[3], PathCounters& pc) {
    pt_rng rng;
    pt_rng_init(&rng, seed, px + py * width, s, 0);
    float x = (float)px + pt_rng_normal(&rng) * 0.5f;                                     // :77
    float y = (float)py + pt_rng_normal(&rng) * 0.5f;                                     // :78
    Ray ray = cam.generate_ray(x, y);                                                     // :80
    const float wavelength_width = MAX_WAVELENGTH - MIN_WAVELENGTH;                       // :82
    float hero = pt_rng_uniform(&rng, 0.0f, pt_uniform_scale(0.0f, wavelength_width / (float)NW));  // :83-84
    float wl[NW];
    for (int i = 0; i < NW; ++i) wl[i] = MIN_WAVELENGTH + (hero + wavelength_width * ((float)i / (float)NW));  // :85-89
    float L[NW];
    radiance(sc, ray, wl, &rng, L, pc);                                                   // :91
    for (int i = 0; i < NW; ++i) {                                                        // :93-97
        float cmf[3];
        from_wavelength(wl[i], cmf);
        for (int c = 0; c < 3; ++c) acc[c] += cmf[c] * L[i] * (1.0f / (float)NW);
    }
}

template <class F> void parallel_for(uint64_t n, int nthreads, F f) {
    if (nthreads <= 1 || n < 2) { f(0, 0, n); return; }
    std::vector<std::thread> th;
    std::atomic<uint64_t> next{0};
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(4096, n / (uint64_t)(nthreads * 8) + 1));
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            for (;;) {
                uint64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                f(t, b, std::min(n, b + chunk));
            }
        });
    for (auto& x : th) x.join();
}

}  // namespace

// =============================================================================================
// C exports for the tests (ctypes)
// =============================================================================================
extern "C" {

struct pto_scene { Scene s; };

// Triangle::new + Triangle::intersect on one ray: returns 1 for Some and fills out[6] =
// {distance, u, v, normal.xyz}.
int pto_triangle_intersect(const float a[3], const float b[3], const float c[3], const float o[3],
                           const float d[3], float max_distance, float out[6]) {
    pt_triangle t{};
    for (int i = 0; i < 3; ++i) { t.a[i] = a[i]; t.b[i] = b[i]; t.c[i] = c[i]; }
    Triangle tr = triangle_new(t);
    Intersection is;
    if (!triangle_intersect(tr, Ray(v3(o), v3(d)), max_distance, &is)) return 0;
    out[0] = is.distance; out[1] = is.u; out[2] = is.v;
    out[3] = is.normal.x; out[4] = is.normal.y; out[5] = is.normal.z;
    return 1;
}

// Triangle::new: out[12] = {n.xyz, d, n1.xyz, d1, n2.xyz, d2}
void pto_triangle_precompute(const pt_triangle* t, float out[12]) {
    Triangle r = triangle_new(*t);
    out[0] = r.n.x; out[1] = r.n.y; out[2] = r.n.z; out[3] = r.d;
    out[4] = r.n1.x; out[5] = r.n1.y; out[6] = r.n1.z; out[7] = r.d1;
    out[8] = r.n2.x; out[9] = r.n2.y; out[10] = r.n2.z; out[11] = r.d2;
}

// Ray::intersects_aabb: returns 1 and *tmin for Some
int pto_ray_aabb(const float o[3], const float d[3], const float bmin[3], const float bmax[3], float* tmin) {
    return Ray(v3(o), v3(d)).intersects_aabb(AABB{v3(bmin), v3(bmax)}, tmin) ? 1 : 0;
}

void pto_camera_ray(uint32_t w, uint32_t h, float sensor_width, float focal, float x, float y, float dir[3]) {
    Ray r = Camera((float)w, (float)h, sensor_width, focal).generate_ray(x, y);
    dir[0] = r.direction.x; dir[1] = r.direction.y; dir[2] = r.direction.z;
}
void pto_from_wavelength(float wl, float out[3]) { from_wavelength(wl, out); }
float pto_reflectance_at_one(const float rgb[3], float wl) { return reflectance_at_one(v3(rgb), wl); }
float pto_d65(float wl) { return d65_sample_one(wl); }
void pto_srgb_to_xyz(const float rgb[3], float out[3]) { srgb_to_xyz(rgb, out); }

// InitialisedSky::from_sky (scene.rs:105-129); camera_transform column-major or NULL
int pto_sky_hosek_wilkie(float turbidity, float albedo, const float zenith[3], const float sun[3],
                         const float* m, pt_sky* out) {
    auto xf = [&](V3 v) {                 // nalgebra Matrix4::transform_vector
        if (!m) return v;
        float n = (m[3] * v.x + m[7] * v.y) + m[11] * v.z;   // bottom row . v
        if (n != 0.0f) v = v / n;
        return V3{(m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
                  (m[2] * v.x + m[6] * v.y) + m[10] * v.z};
    };
    V3 z = normalize(xf(v3(zenith)));
    V3 s = normalize(xf(v3(sun)));
    float elev = std::asin(dot(z, s));
    if (!(elev > 0.0f)) return PT_ERR_INVALID;                                            // scene.rs:121
    if (!(turbidity >= 1.0f && turbidity <= 10.0f)) return PT_ERR_INVALID;                // hosek_wilkie.rs:155,351-352
    HosekWilkie hw = hw_new(elev, turbidity, albedo);
    std::memset(out, 0, sizeof(*out));
    out->kind = PT_SKY_HOSEK_WILKIE;
    std::memcpy(out->configs, hw.configs, sizeof(hw.configs));
    std::memcpy(out->radiances, hw.radiances, sizeof(hw.radiances));
    out->turbidity = hw.turbidity; out->solar_radius = hw.solar_radius;
    out->zenith[0] = z.x; out->zenith[1] = z.y; out->zenith[2] = z.z;
    out->sun[0] = s.x; out->sun[1] = s.y; out->sun[2] = s.z;
    return PT_OK;
}

// Scene: Triangle::new on every triangle, BVH::build unless desc->nodes is given.
int pto_scene_create(const pt_scene_desc* desc, pto_scene** out) {
    auto sc = std::make_unique<pto_scene>();
    Scene& s = sc->s;
    s.triangles.reserve(desc->n_triangles);
    for (uint32_t i = 0; i < desc->n_triangles; ++i) {
        if (desc->triangles[i].material_index >= desc->n_materials) return PT_ERR_INVALID;
        s.triangles.push_back(triangle_new(desc->triangles[i]));
    }
    s.materials.assign(desc->materials, desc->materials + desc->n_materials);
    for (uint32_t i = 0; i < desc->n_images; ++i) {
        const pt_image& im = desc->images[i];
        s.images.push_back(Image{im.width, im.height, im.channels,
                                 std::vector<float>(im.texels, im.texels + (size_t)im.width * im.height * im.channels)});
    }
    s.sky = desc->sky;
    std::memcpy(s.hw.configs, desc->sky.configs, sizeof(s.hw.configs));
    std::memcpy(s.hw.radiances, desc->sky.radiances, sizeof(s.hw.radiances));
    s.hw.turbidity = desc->sky.turbidity; s.hw.solar_radius = desc->sky.solar_radius;
    if (desc->nodes && desc->n_nodes) {
        s.nodes.reserve(desc->n_nodes);
        for (uint32_t i = 0; i < desc->n_nodes; ++i) {
            const pt_bvh_node& n = desc->nodes[i];
            s.nodes.push_back(Node{n.is_leaf != 0, n.shape_index, n.child_l_index, n.child_r_index,
                                   AABB{v3(n.child_l_aabb.min), v3(n.child_l_aabb.max)},
                                   AABB{v3(n.child_r_aabb.min), v3(n.child_r_aabb.max)}});
        }
    } else if (!s.triangles.empty()) {                                                    // bvh.rs:346-356
        std::vector<uint32_t> indices(s.triangles.size());
        for (uint32_t i = 0; i < indices.size(); ++i) indices[i] = i;
        s.nodes.reserve(s.triangles.size() * 2);
        try { build_rec(s.triangles, indices, s.nodes); }
        catch (const BuildError&) { return PT_ERR_BVH_DEGENERATE; }
    }
    *out = sc.release();
    return PT_OK;
}
void pto_scene_destroy(pto_scene* s) { delete s; }
uint32_t pto_scene_n_nodes(const pto_scene* s) { return (uint32_t)s->s.nodes.size(); }
void pto_scene_get_nodes(const pto_scene* s, pt_bvh_node* out) {
    for (size_t i = 0; i < s->s.nodes.size(); ++i) {
        const Node& n = s->s.nodes[i];
        pt_bvh_node o{};
        o.is_leaf = n.is_leaf; o.shape_index = n.shape_index;
        o.child_l_index = n.child_l_index; o.child_r_index = n.child_r_index;
        if (!n.is_leaf) {
            for (int k = 0; k < 3; ++k) {
                o.child_l_aabb.min[k] = n.child_l_aabb.min[k]; o.child_l_aabb.max[k] = n.child_l_aabb.max[k];
                o.child_r_aabb.min[k] = n.child_r_aabb.min[k]; o.child_r_aabb.max[k] = n.child_r_aabb.max[k];
            }
        }
        out[i] = o;
    }
}

// BVH::traverse on a ray batch.  counters[2] (nullable) += {internal nodes visited, triangles
// tested} under the reference's own traversal rule — the n_node / n_tri of SURVEY.md §8(d).
void pto_trace(const pto_scene* sc, const pt_ray* rays, const uint32_t* start, uint64_t n,
               pt_hit* out, uint64_t* counters, int nthreads) {
    const Scene& s = sc->s;
    std::vector<Counters> cnt((size_t)std::max(1, nthreads));
    parallel_for(n, nthreads, [&](int t, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            Ray ray(v3(rays[i].o), v3(rays[i].d));
            TravResult r{false, PT_NO_TRIANGLE, {}};
            if (!s.nodes.empty())
                r = traverse_rec(s.nodes, 0, ray, start ? start[i] : PT_NO_TRIANGLE, s.triangles, INF, cnt[t]);
            if (r.some) out[i] = pt_hit{r.tri, r.isect.distance, r.isect.u, r.isect.v};
            else out[i] = pt_hit{PT_NO_TRIANGLE, 0.0f, 0.0f, 0.0f};
        }
    });
    if (counters) for (auto& c : cnt) { counters[0] += c.nodes; counters[1] += c.tris; }
}

// pto_trace, and per ray whether any triangle test on its way took the sign of a NaN (see tl_nan_sign_test): on those rays
// the reference's answer is platform-dependent.
void pto_trace_nan_flags(const pto_scene* sc, const pt_ray* rays, const uint32_t* start, uint64_t n,
                         pt_hit* out, uint8_t* nan_flags, int nthreads) {
    const Scene& s = sc->s;
    parallel_for(n, nthreads, [&](int, uint64_t b, uint64_t e) {
        Counters cnt;
        for (uint64_t i = b; i < e; ++i) {
            Ray ray(v3(rays[i].o), v3(rays[i].d));
            TravResult r{false, PT_NO_TRIANGLE, {}};
            tl_nan_sign_test = false;
            if (!s.nodes.empty())
                r = traverse_rec(s.nodes, 0, ray, start ? start[i] : PT_NO_TRIANGLE, s.triangles, INF, cnt);
            nan_flags[i] = tl_nan_sign_test ? 1 : 0;
            if (r.some) out[i] = pt_hit{r.tri, r.isect.distance, r.isect.u, r.isect.v};
            else out[i] = pt_hit{PT_NO_TRIANGLE, 0.0f, 0.0f, 0.0f};
        }
    });
}

// Second oracle: no BVH, every triangle tested in index order, minimum distance wins
// (first index on ties).  Independent of build and traversal order.
void pto_trace_brute(const pto_scene* sc, const pt_ray* rays, const uint32_t* start, uint64_t n,
                     pt_hit* out, int nthreads) {
    const Scene& s = sc->s;
    parallel_for(n, nthreads, [&](int, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            Ray ray(v3(rays[i].o), v3(rays[i].d));
            pt_hit best{PT_NO_TRIANGLE, 0, 0, 0};
            float bt = INF;
            uint32_t st = start ? start[i] : PT_NO_TRIANGLE;
            for (uint32_t k = 0; k < s.triangles.size(); ++k) {
                if (k == st) continue;
                Intersection is;
                if (triangle_intersect(s.triangles[k], ray, INF, &is) && is.distance < bt) {
                    bt = is.distance; best = pt_hit{k, is.distance, is.u, is.v};
                }
            }
            out[i] = best;
        }
    });
}

// Design aid: visit counts of an iterative near-first traversal with a GLOBAL best distance
// (enter a child iff its entry distance < best; later hit wins ties).  counters[4] +=
// {nodes, tris, sum of max stack depth, max stack depth}.
void pto_trace_globalbest_stats(const pto_scene* sc, const pt_ray* rays, uint64_t n, pt_hit* out,
                                uint64_t* counters, uint32_t* per_ray /* nullable: {nodes, tris, pops} per ray */) {
    const Scene& s = sc->s;
    uint64_t nodes = 0, tris = 0, sumdepth = 0, maxdepth = 0;
    std::vector<std::pair<uint32_t, float>> stack;
    for (uint64_t i = 0; i < n; ++i) {
        Ray ray(v3(rays[i].o), v3(rays[i].d));
        pt_hit best{PT_NO_TRIANGLE, 0, 0, 0};
        float bt = INF;
        stack.clear();
        size_t md = 0;
        const uint64_t nodes0 = nodes, tris0 = tris;
        uint32_t pops = 0;
        if (!s.nodes.empty()) stack.push_back({0u, 0.0f});
        while (!stack.empty()) {
            auto [ni, tmin] = stack.back(); stack.pop_back();
            pops++;
            if (!(bt > tmin)) continue;
            for (;;) {
                const Node& nd = s.nodes[ni];
                if (nd.is_leaf) {
                    tris++;
                    Intersection is;
                    if (triangle_intersect(s.triangles[nd.shape_index], ray, INF, &is) && is.distance <= bt) {
                        bt = is.distance; best = pt_hit{nd.shape_index, is.distance, is.u, is.v};
                    }
                    break;
                }
                nodes++;
                float lm = 0, rm = 0;
                bool l = ray.intersects_aabb(nd.child_l_aabb, &lm) && bt > lm;
                bool r = ray.intersects_aabb(nd.child_r_aabb, &rm) && bt > rm;
                if (l && r) {
                    if (lm < rm) { stack.push_back({nd.child_r_index, rm}); ni = nd.child_l_index; }
                    else         { stack.push_back({nd.child_l_index, lm}); ni = nd.child_r_index; }
                    md = std::max(md, stack.size());
                } else if (l) ni = nd.child_l_index;
                else if (r) ni = nd.child_r_index;
                else break;
            }
        }
        if (out) out[i] = best;
        if (per_ray) { per_ray[3 * i] = (uint32_t)(nodes - nodes0); per_ray[3 * i + 1] = (uint32_t)(tris - tris0); per_ray[3 * i + 2] = pops; }
        sumdepth += md; maxdepth = std::max<uint64_t>(maxdepth, md);
    }
    counters[0] += nodes; counters[1] += tris; counters[2] += sumdepth;
    counters[3] = std::max(counters[3], maxdepth);
}

// Texture::sample (material.rs:90-124) on image `image` of the scene: kind = PT_TEX_NEAREST / PT_TEX_BILINEAR
void pto_texture_sample(const pto_scene* sc, uint32_t kind, uint32_t image, float u, float v, float out[3]) {
    pt_texture t{};
    t.kind = kind; t.image = image;
    const V3 c = texture_sample(sc->s, t, u, v);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

// evaluate_sky for one ray direction
void pto_evaluate_sky(const pto_scene* sc, const float d[3], const float wl[8], float out[8]) {
    evaluate_sky(sc->s, Ray({0, 0, 0}, v3(d)), wl, out);
}

// The sample loop main.rs:64-109 for samples [spp_begin, spp_begin+spp_count).
// mode: PT_RENDER_FINAL / PT_RENDER_SUM as ptcore.h.  counters[3] (nullable) +=
// {paths, segments(traversals), internal nodes visited}; counters[3] += triangles tested.
void pto_render(const pto_scene* sc, const pt_camera* cam, uint32_t spp_begin, uint32_t spp_count,
                uint32_t spp_total, uint64_t seed, uint32_t mode, float* xyzw, uint64_t* counters,
                int nthreads) {
    const Scene& s = sc->s;
    const uint32_t W = cam->width, H = cam->height;
    Camera camera((float)W, (float)H, cam->sensor_width, cam->focal_length);
    // work items as main.rs:64-68: (x, 32-row strip)
    const uint32_t TILE = 32;
    const uint64_t strips = (uint64_t)((H + TILE - 1) / TILE) * W;
    std::vector<PathCounters> pcs((size_t)std::max(1, nthreads));
    for (auto& p : pcs) p.normalize_sky = (mode & PT_RENDER_NORMALIZE_SKY_DIRECTION) != 0u;
    mode &= ~PT_RENDER_NORMALIZE_SKY_DIRECTION;
    parallel_for(strips, nthreads, [&](int t, uint64_t b, uint64_t e) {
        for (uint64_t item = b; item < e; ++item) {
            uint32_t x = (uint32_t)(item % W);
            uint32_t y0 = (uint32_t)(item / W) * TILE;
            uint32_t th = std::min(H - y0, TILE);
            for (uint32_t y = y0; y < y0 + th; ++y) {
                float acc[3] = {0, 0, 0};
                for (uint32_t sidx = spp_begin; sidx < spp_begin + spp_count; ++sidx)
                    sample_pixel(s, camera, x, y, W, sidx, seed, acc, pcs[t]);
                float* px = &xyzw[4 * ((size_t)x + (size_t)y * W)];
                if (mode == PT_RENDER_FINAL) {
                    float inv = 1.0f / (float)spp_total;                                  // main.rs:100
                    px[0] = acc[0] * inv; px[1] = acc[1] * inv; px[2] = acc[2] * inv; px[3] = 1.0f;
                } else {
                    px[0] = acc[0]; px[1] = acc[1]; px[2] = acc[2]; px[3] = (float)spp_count;
                }
            }
        }
    });
    if (counters) {
        for (auto& p : pcs) {
            counters[0] += (uint64_t)0; counters[1] += p.segments; counters[2] += p.trav.nodes; counters[3] += p.trav.tris;
        }
        counters[0] += (uint64_t)W * H * spp_count;
    }
}

// shared-math probes so tests can pin pt_math.h / pt_rng.h against float64 references
float pto_math(int fn, float x) {
    switch (fn) {
        case 0: return pt_expf(x);
        case 1: return pt_logf(x);
        case 2: return pt_acosf(x);
        case 3: return pt_sinf(x);
        case 4: return pt_cbrtf(x);
        case 5: return pt_pow1p5(x);
        case 6: return pt_norm_ppf(x);
        case 7: return pt_asinf(x);
    }
    return 0.0f;
}
void pto_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    pt_u32x4 c; for (int i = 0; i < 4; ++i) c.v[i] = ctr[i];
    pt_u32x4 r = pt_philox4x32_10(c, key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
// n draws of stream (seed, pixel, sample): kind 0 u32 (as float bits via memcpy), 1 f32, 2 normal,
// 3 uniform(lo,hi), 4 unit sphere (3 floats per draw)
void pto_rng_draw(uint64_t seed, uint32_t pixel, uint32_t sample, int kind, float lo, float hi,
                  uint32_t n, float* out, uint32_t* index_out) {
    pt_rng r; pt_rng_init(&r, seed, pixel, sample, 0);
    float scale = (kind == 3) ? pt_uniform_scale(lo, hi) : 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        switch (kind) {
            case 0: { uint32_t u = pt_rng_u32(&r); std::memcpy(&out[i], &u, 4); break; }
            case 1: out[i] = pt_rng_f32(&r); break;
            case 2: out[i] = pt_rng_normal(&r); break;
            case 3: out[i] = pt_rng_uniform(&r, lo, scale); break;
            case 4: pt_rng_unit_sphere(&r, &out[3 * i]); break;
        }
    }
    if (index_out) *index_out = r.index;
}

int pto_uses_libm(void) {
#ifdef PTO_USE_LIBM
    return 1;
#else
    return 0;
#endif
}

}  // extern "C"
