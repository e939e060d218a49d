// scoped_model.cpp — host-side model of the default closest-hit kernel (PT_TRAVERSE_EXACT, trace_kernels.h:
// k_trace_scoped_persistent).  TEST INFRASTRUCTURE: built into tests/cpp/libscopedmodel.so, loaded by
// tests/test_scoped_model.py only.
//
// The kernel and this model share the per-record decision code (pathtracer_amd/csrc/trace_step.h: slab arithmetic,
// derived child boxes, near / far order, the three stack entries) and the triangle test; what differs is only the
// container of the stack (LDS ring over an HBM column there, std::vector here) and who hands out rays.  The model lets
// the CPU suite check, without a GPU, that the SCOPED traversal - two tree levels per step, one running minimum per
// pending sibling instead of the recursion's frames - returns what traverse_recursive (bvh.rs:250-338) returns, ray by
// ray, on scenes built to provoke ties, rounding gaps, flat boxes and duplicate triangles; and that the children's boxes
// derived inside ptc_wide_step equal the direct Ray::intersects_aabb of the binary node's stored boxes.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/ptcore.h"
#include "../../pathtracer_amd/csrc/trace_step.h"

namespace {

struct F4 { float x, y, z, w; };

inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

inline float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// Triangle::new (triangle.rs:50-60), as pt_scene_create precomputes it
void precompute(const pt_triangle& t, F4 rec[3]) {
    float ab[3], ac[3], n[3], n1[3], n2[3], tmp[3];
    for (int k = 0; k < 3; ++k) { ab[k] = t.b[k] - t.a[k]; ac[k] = t.c[k] - t.a[k]; }
    cross3(ab, ac, n);
    const float dd = dot3(t.a, n), nn = dot3(n, n);
    cross3(ac, n, tmp);
    for (int k = 0; k < 3; ++k) n1[k] = tmp[k] / nn;
    const float d1 = -dot3(t.a, n1);
    cross3(n, ab, tmp);
    for (int k = 0; k < 3; ++k) n2[k] = tmp[k] / nn;
    const float d2 = -dot3(t.a, n2);
    rec[0] = {n[0], n[1], n[2], dd};
    rec[1] = {n1[0], n1[1], n1[2], d1};
    rec[2] = {n2[0], n2[1], n2[2], d2};
}

struct Wide { F4 lox, loy, loz, hix, hiy, hiz, rw; };

// the wide record of binary node `i` (fixed slots: 0,1 = right child's right / left, 2,3 = left child's right / left;
// a leaf child in the first slot of its pair, NaN box in the other) - refs to internal nodes are indices into `nodes`
Wide make_wide(const pt_bvh_node* nodes, uint32_t i) {
    const float nan = std::numeric_limits<float>::quiet_NaN();
    float lo[3][4], hi[3][4];
    uint32_t ref[4] = {PTC_IDLE, PTC_IDLE, PTC_IDLE, PTC_IDLE};
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 4; ++k) { lo[a][k] = nan; hi[a][k] = nan; }
    auto put = [&](int slot, const pt_aabb& b, uint32_t child) {
        for (int a = 0; a < 3; ++a) { lo[a][slot] = b.min[a]; hi[a][slot] = b.max[a]; }
        ref[slot] = nodes[child].is_leaf ? (PTC_LEAF_BIT | nodes[child].shape_index) : child;
    };
    const pt_bvh_node& x = nodes[i];
    for (int side = 1; side >= 0; --side) {
        const uint32_t c = side ? x.child_r_index : x.child_l_index;
        const int base = side ? 0 : 2;
        if (nodes[c].is_leaf) put(base, side ? x.child_r_aabb : x.child_l_aabb, c);
        else { put(base, nodes[c].child_r_aabb, nodes[c].child_r_index); put(base + 1, nodes[c].child_l_aabb, nodes[c].child_l_index); }
    }
    Wide w;
    w.lox = {lo[0][0], lo[0][1], lo[0][2], lo[0][3]}; w.loy = {lo[1][0], lo[1][1], lo[1][2], lo[1][3]}; w.loz = {lo[2][0], lo[2][1], lo[2][2], lo[2][3]};
    w.hix = {hi[0][0], hi[0][1], hi[0][2], hi[0][3]}; w.hiy = {hi[1][0], hi[1][1], hi[1][2], hi[1][3]}; w.hiz = {hi[2][0], hi[2][1], hi[2][2], hi[2][3]};
    w.rw = {u2f(ref[0]), u2f(ref[1]), u2f(ref[2]), u2f(ref[3])};
    return w;
}

struct Entry { uint32_t ref; float e; };

// optional per-trip log for tools/wave_sim.py: one byte per loop trip of every ray (bit 0: a record was visited, bit 1: a
// triangle was tested or skipped as `start`, bits 2..7: entries popped after it), rays separated by 0xFF
std::vector<uint8_t>* g_trip_log = nullptr;

}  // namespace

extern "C" {

// out[i] = the scoped traversal's hit of ray i; flagged[i] = 1 where the kernel hands the ray to the replay kernel instead
// (irregular ray, or a triangle hit at a non-finite distance) - out[i] is then meaningless.
// stats: [0] records whose derived child boxes differ from the direct slab test of the node's stored boxes, or whose
//            sign-select form decides differently from the general form (must stay 0),
//        [1] deepest stack, [2] wide records visited, [3] triangle tests, [4] RESTORE entries pushed,
//        [5] entries pushed (RESTOREs included), [6] popped entries that were entered, [7] loop trips (record + leaf + pops of one lane)
// per_ray_trips (may be NULL): loop trips of each ray - the length of its dependent chain of record / triangle fetches
int ptm_trace_scoped_trips(const pt_bvh_node* nodes, uint32_t n_nodes, const pt_triangle* tris, uint32_t n_tris,
                           const pt_ray* rays, const uint32_t* start, uint64_t n, pt_hit* out, uint8_t* flagged, uint64_t stats[8],
                           uint32_t* per_ray_trips) {
    std::vector<F4> rec((size_t)n_tris * 3);
    for (uint32_t i = 0; i < n_tris; ++i) precompute(tris[i], &rec[(size_t)i * 3]);
    std::vector<Wide> wide(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) if (!nodes[i].is_leaf) wide[i] = make_wide(nodes, i);
    for (int k = 0; k < 8; ++k) stats[k] = 0;
    const float INF = std::numeric_limits<float>::infinity();
    std::vector<Entry> st;
    for (uint64_t i = 0; i < n; ++i) {
        flagged[i] = 0;
        out[i] = pt_hit{PT_NO_TRIANGLE, 0.0f, 0.0f, 0.0f};
        if (n_nodes == 0) continue;
        const F4 o{rays[i].o[0], rays[i].o[1], rays[i].o[2], 0.0f}, d{rays[i].d[0], rays[i].d[1], rays[i].d[2], 0.0f};
        const DRay r = ptc_make_ray(o, d);
        if (!ptc_ray_is_regular(r)) { flagged[i] = 1; continue; }
        const bool sign_ok = ptc_ray_sign_select(r);
        const uint32_t skip = start ? start[i] : PT_NO_TRIANGLE;
        float best_t = INF, best_u = 0.0f, best_v = 0.0f, m_cur = INF;
        uint32_t best_id = PT_NO_TRIANGLE;
        uint32_t cur = nodes[0].is_leaf ? (PTC_LEAF_BIT | nodes[0].shape_index) : 0u;
        st.clear();
        auto push = [&](uint32_t ref, float e) {
            if (m_cur != INF) { st.push_back({PTC_RESTORE, m_cur}); m_cur = INF; stats[4] += 1; stats[5] += 1; }
            st.push_back({ref, e});
            stats[5] += 1;
            if (st.size() > stats[1]) stats[1] = st.size();
        };
        bool alive = true;
        const uint64_t trips_before = stats[7];
        while (alive) {
            stats[7] += 1;
            uint32_t log_bits = 0u, log_pops = 0u;
            if (!(cur & PTC_LEAF_BIT)) {
                log_bits |= 1u;
                const Wide& w = wide[cur];
                stats[2] += 1;
                PtcWideStep s;
                ptc_wide_step(r, w.lox, w.loy, w.loz, w.hix, w.hiy, w.hiz, w.rw, s);
                if (sign_ok) {   // the sign-select form the kernel uses for such rays: same decisions, bit for bit
                    PtcWideStep f;
                    ptc_wide_step_sign(r, r.ix < 0.0f ? w.hix : w.lox, r.iy < 0.0f ? w.hiy : w.loy, r.iz < 0.0f ? w.hiz : w.loz,
                                       r.ix < 0.0f ? w.lox : w.hix, r.iy < 0.0f ? w.loy : w.hiy, r.iz < 0.0f ? w.loz : w.hiz, w.rw, f);
                    const bool same = f.cur == s.cur && f.v0 == s.v0 && f.v1 == s.v1 && f.v2 == s.v2 &&
                                      (!s.v0 || (f.ref0 == s.ref0 && f.e0 == s.e0)) && (!s.v1 || (f.ref1 == s.ref1 && f.e1 == s.e1)) &&
                                      (!s.v2 || (f.ref2 == s.ref2 && f.e2 == s.e2));
                    if (!same) stats[0] += 1;
                }
                {   // the derived children's boxes against the direct test of the stored ones (bvh.rs:260-261)
                    const pt_bvh_node& x = nodes[cur];
                    float lt = 0.0f, rt = 0.0f;
                    const bool lh = ptc_slab(r, x.child_l_aabb.min[0], x.child_l_aabb.min[1], x.child_l_aabb.min[2], x.child_l_aabb.max[0], x.child_l_aabb.max[1], x.child_l_aabb.max[2], lt);
                    const bool rh = ptc_slab(r, x.child_r_aabb.min[0], x.child_r_aabb.min[1], x.child_r_aabb.min[2], x.child_r_aabb.max[0], x.child_r_aabb.max[1], x.child_r_aabb.max[2], rt);
                    bool ok = lh == s.hit_l && rh == s.hit_r;
                    if (lh && !(lt == s.entry_l)) ok = false;          // (== : the sign of a zero may differ)
                    if (rh && !(rt == s.entry_r)) ok = false;
                    if (!ok) stats[0] += 1;
                }
                if (s.v0) push(s.ref0, s.e0);
                if (s.v1) push(s.ref1, s.e1);
                if (s.v2) push(s.ref2, s.e2);
                cur = s.cur;
            }
            if ((cur & PTC_LEAF_BIT) && cur != PTC_IDLE) {
                log_bits |= 2u;
                const uint32_t tri = cur & PTC_INDEX_MASK;
                if (tri != skip) {
                    stats[3] += 1;
                    float dist, u, v;
                    if (ptc_tri_test(r, rec[(size_t)tri * 3], rec[(size_t)tri * 3 + 1], rec[(size_t)tri * 3 + 2], dist, u, v)) {
                        if (!(std::fabs(dist) < INF)) { flagged[i] = 1; alive = false; break; }
                        if (dist <= best_t) { best_t = dist; best_u = u; best_v = v; best_id = tri; }
                        m_cur = dist < m_cur ? dist : m_cur;
                    }
                }
                cur = PTC_IDLE;
            }
            while (cur == PTC_IDLE) {
                if (st.empty()) { alive = false; break; }
                const Entry e = st.back();
                st.pop_back();
                log_pops += 1u;
                if (e.ref == PTC_RESTORE) { m_cur = e.e < m_cur ? e.e : m_cur; continue; }
                const bool visit = m_cur > e.e;                      // first_distance > second_child_min (bvh.rs:285)
                if (e.ref & PTC_COMPANION) {
                    const Entry c = st.back();                       // the far child's other grandchild
                    st.pop_back();
                    if (visit) push(c.ref, c.e);
                }
                if (visit) { cur = e.ref & ~PTC_COMPANION; stats[6] += 1; }
            }
            if (g_trip_log) g_trip_log->push_back((uint8_t)(log_bits | ((log_pops < 62u ? log_pops : 62u) << 2)));
        }
        if (g_trip_log) g_trip_log->push_back(0xFF);
        if (!flagged[i]) out[i] = pt_hit{best_id, best_id == PT_NO_TRIANGLE ? 0.0f : best_t, best_u, best_v};
        if (per_ray_trips) per_ray_trips[i] = (uint32_t)(stats[7] - trips_before);
    }
    return 0;
}

// trip log of the rays of one call (see g_trip_log): returns its length; `log` may be NULL to ask for the size of the last one
uint64_t ptm_trace_scoped_log(const pt_bvh_node* nodes, uint32_t n_nodes, const pt_triangle* tris, uint32_t n_tris,
                              const pt_ray* rays, const uint32_t* start, uint64_t n, uint8_t* log, uint64_t capacity) {
    static std::vector<uint8_t> last;
    if (rays) {
        last.clear();
        g_trip_log = &last;
        std::vector<pt_hit> out(n);
        std::vector<uint8_t> flagged(n);
        uint64_t stats[8];
        ptm_trace_scoped_trips(nodes, n_nodes, tris, n_tris, rays, start, n, out.data(), flagged.data(), stats, nullptr);
        g_trip_log = nullptr;
    }
    if (log && capacity >= last.size()) std::memcpy(log, last.data(), last.size());
    return last.size();
}

int ptm_trace_scoped(const pt_bvh_node* nodes, uint32_t n_nodes, const pt_triangle* tris, uint32_t n_tris,
                     const pt_ray* rays, const uint32_t* start, uint64_t n, pt_hit* out, uint8_t* flagged, uint64_t stats[8]) {
    return ptm_trace_scoped_trips(nodes, n_nodes, tris, n_tris, rays, start, n, out, flagged, stats, nullptr);
}

}  // extern "C"
