// trace_step.h — the per-lane arithmetic of the closest-hit kernels, shared by the device code (trace.hip) and by the
// host-side model of the traversal that the CPU tests walk (tests/cpp/scoped_model.cpp): ray setup, the slab test, the
// Havel–Herout triangle test, and the decision a lane takes at one 4-wide record under the RECURSION's rules.
//
//   Ray::new / Ray::intersects_aabb        bvh.rs:380-410
//   Triangle::intersect                    triangle.rs:82-115
//   BVHNode::traverse_recursive            bvh.rs:259-325 (the both / one / none cases and the near-first order)
//
// Everything here is written in the reference's operation order; translation units that include it are built with
// -ffp-contract=off (no FMA), IEEE division, denormals kept.
#ifndef PTC_TRACE_STEP_H
#define PTC_TRACE_STEP_H

#include <stdint.h>

#include "../../include/pt_math.h"

#ifndef PTC_LEAF_BIT
#define PTC_LEAF_BIT  0x80000000u
#define PTC_FRAME_BIT 0x40000000u
#define PTC_INDEX_MASK 0x3FFFFFFFu
#endif
#define PTC_IDLE      0xFFFFFFFFu   /* "needs a pop": never a valid reference (triangle indices are < 2^30 - 2)        */
#define PTC_RESTORE   0xFFFFFFFEu   /* scoped stack: {RESTORE, bits(m)} folds m back into the running scope minimum    */
#define PTC_COMPANION PTC_FRAME_BIT /* scoped stack: the entry directly below belongs to the same (far) child          */

struct DRay { float ox, oy, oz, dx, dy, dz, ix, iy, iz; };

template <class F4>
PT_HD DRay ptc_make_ray(F4 o, F4 d) {
    DRay r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.ix = 1.0f / d.x; r.iy = 1.0f / d.y; r.iz = 1.0f / d.z;      // bvh.rs:384
    return r;
}

// A ray is REGULAR when its origin and direction are finite (zero and denormal direction components included: their
// +-inf inverse is part of the reference's arithmetic, bvh.rs:384).  ptc_wide_step's derived child boxes are exact for
// such rays (see there); a ray with an infinite or NaN component is handed to the frame-by-frame replay of the recursion.
PT_HD bool ptc_ray_is_regular(const DRay& r) {
    const float big = pt_inf();
    return pt_fabs(r.ox) < big && pt_fabs(r.oy) < big && pt_fabs(r.oz) < big &&
           pt_fabs(r.dx) < big && pt_fabs(r.dy) < big && pt_fabs(r.dz) < big;
}

// Ray::intersects_aabb (bvh.rs:390-410); returns hit flag, entry distance in tmin
PT_HD bool ptc_slab(const DRay& r, float bx0, float by0, float bz0, float bx1, float by1, float bz1, float& tmin_out) {
    float t1x = (bx0 - r.ox) * r.ix, t2x = (bx1 - r.ox) * r.ix;
    float t1y = (by0 - r.oy) * r.iy, t2y = (by1 - r.oy) * r.iy;
    float t1z = (bz0 - r.oz) * r.iz, t2z = (bz1 - r.oz) * r.iz;
    float tmin = pt_fmin(t1x, t2x);
    float tmax = pt_fmax(t1x, t2x);
    tmin = pt_fmax(tmin, pt_fmin(t1y, t2y));
    tmax = pt_fmin(tmax, pt_fmax(t1y, t2y));
    tmin = pt_fmax(tmin, pt_fmin(t1z, t2z));
    tmax = pt_fmin(tmax, pt_fmax(t1z, t2z));
    tmin = pt_fmax(tmin, 0.0f);
    tmin_out = tmin;
    return tmax >= tmin;
}

// Two boxes per packed instruction on the device (v_pk_add_f32 / v_pk_mul_f32 round each half like the scalar forms).
#if defined(__clang__)
typedef float ptc_f2 __attribute__((ext_vector_type(2)));
#else
struct ptc_f2 { float x, y; };
inline ptc_f2 operator-(ptc_f2 a, ptc_f2 b) { return {a.x - b.x, a.y - b.y}; }
inline ptc_f2 operator*(ptc_f2 a, ptc_f2 b) { return {a.x * b.x, a.y * b.y}; }
#endif
struct PtcPair { ptc_f2 t1, t2; };
PT_HD PtcPair ptc_slab_axis2(float lo_a, float lo_b, float hi_a, float hi_b, float o, float inv) {
    const ptc_f2 lo = {lo_a, lo_b}, hi = {hi_a, hi_b}, oo = {o, o}, ii = {inv, inv};
    PtcPair p;
    p.t1 = (lo - oo) * ii;
    p.t2 = (hi - oo) * ii;
    return p;
}
// the min/max chain of bvh.rs:394-405 on per-axis (near, far) = (min(t1,t2), max(t1,t2))
PT_HD bool ptc_slab_close_nf(float nx, float fx, float ny, float fy, float nz, float fz, float& tmin_out) {
    float tmin = nx;
    float tmax = fx;
    tmin = pt_fmax(tmin, ny);
    tmax = pt_fmin(tmax, fy);
    tmin = pt_fmax(tmin, nz);
    tmax = pt_fmin(tmax, fz);
    tmin = pt_fmax(tmin, 0.0f);
    tmin_out = tmin;
    return tmax >= tmin;
}
PT_HD bool ptc_slab_close(float t1x, float t2x, float t1y, float t2y, float t1z, float t2z, float& tmin_out) {
    return ptc_slab_close_nf(pt_fmin(t1x, t2x), pt_fmax(t1x, t2x), pt_fmin(t1y, t2y), pt_fmax(t1y, t2y),
                             pt_fmin(t1z, t2z), pt_fmax(t1z, t2z), tmin_out);
}

// Triangle::intersect with max_distance = INFINITY (bvh.rs:332), branch-free until the verdict
template <class F4>
PT_HD bool ptc_tri_test(const DRay& r, F4 t0, F4 t1, F4 t2, float& dist, float& uo, float& vo) {
    float det = (r.dx * t0.x + r.dy * t0.y) + r.dz * t0.z;                   // triangle.rs:84
    float t = t0.w - ((r.ox * t0.x + r.oy * t0.y) + r.oz * t0.z);            // :89
    float m = pt_inf() * det - t;                                            // :91
    float px = r.ox * det + r.dx * t;                                        // :95
    float py = r.oy * det + r.dy * t;
    float pz = r.oz * det + r.dz * t;
    float u = ((px * t1.x + py * t1.y) + pz * t1.z) + det * t1.w;            // :96
    float v = ((px * t2.x + py * t2.y) + pz * t2.z) + det * t2.w;            // :101
    float du = det - u;
    float duv = du - v;                                                      // (det - u) - v, :102
    uint32_t bad = (pt_f2u(m) ^ pt_f2u(t)) | (pt_f2u(du) ^ pt_f2u(u)) | (pt_f2u(duv) ^ pt_f2u(v));
    float inv = 1.0f / det;                                                  // :106
    float d = t * inv;                                                       // :108
    bool ok = (det != 0.0f) && ((bad >> 31) == 0u);
    // relative_eq!(distance, 0.0): equal, or |distance| <= f32::EPSILON      :110
    ok = ok && !(d == 0.0f || pt_fabs(d) <= 1.1920928955078125e-7f);
    dist = d;
    uo = u * inv;
    vo = v * inv;
    return ok;
}

// ---- one 4-wide record under the recursion's rules ---------------------------------------------------------------------
// A wide record stands for a binary node X and its two children: slots 0, 1 = the RIGHT child's right and left child
// (RR, RL), slots 2, 3 = the LEFT child's (LR, LL); a child that is a leaf sits in the first slot of its pair and the
// other slot holds a NaN box, which no ray hits.  What traverse_recursive does at X and then at the child it enters
// (bvh.rs:259-325) needs the children's own boxes as well; those are not stored: a child's box is the join of its
// children's boxes (AABB::join is an exact min / max, bvh.rs:54-67), and the per-axis (near, far) pair of the joined box
// is exactly (min of the two near values, max of the two far values), bit for bit what Ray::intersects_aabb computes
// from the joined box (up to the sign of a zero, which nothing below observes):
//   * inverse direction finite and non-zero: t = fl(fl(c - o) * inv) is a monotone function of the box coordinate c
//     (rounding is monotone) and never NaN for finite c and o, so min / max commute with it;
//   * inverse direction +-inf (direction component +-0 or denormal): t is +-inf by the sign of c - o, NaN where c == o
//     exactly (denormals are kept, so fl(c - o) == 0 iff c == o); minNum / maxNum drop the NaN.  Going through the six
//     positions of o relative to [lo, hi] (outside below, on lo, inside, on hi, outside above, lo == o == hi) for the
//     joined box and for its two parts gives the same (near, far) either way;
//   * inverse direction 0 (infinite component) or NaN: the ray is not REGULAR and never gets here;
//   * the NaN box of an empty slot contributes nothing (minNum / maxNum), so a leaf child's pair is its own slot's.
// tests/cpp/scoped_model.cpp checks the claim against the direct computation on every record it visits.
//
// The decision:  at X   both children hit: near = left iff l_min < r_min (bvh.rs:266), far waits for the near result;
//                       one hit: that one;  none: nothing.
//                at the child entered: the same over its two children (the slots of its pair).
// Output: `cur` = what to enter now (PTC_IDLE: nothing), and up to three stack entries in push order:
//   p0  the far child's far grandchild      {ref, its own entry distance}             (plain)
//   p1  the far child's near grandchild     {ref | COMPANION iff p0, the FAR CHILD's entry distance}
//       - popped first: the far child is entered iff the near child's result is farther than the far CHILD's box
//         (bvh.rs:285); if it is not, p0 goes with it
//   p2  the near child's far grandchild     {ref, its own entry distance}             (plain)
struct PtcWideStep {
    uint32_t cur;
    bool v0, v1, v2;
    uint32_t ref0, ref1, ref2;
    float e0, e1, e2;
    bool hit_l, hit_r;          // the derived children's boxes, for the model's check against the stored ones
    float entry_l, entry_r;
};

// minNum / maxNum on values that are results of arithmetic (never signalling NaNs).  On the device these are the bare
// v_min_f32 / v_max_f32 (IEEE mode: the non-NaN operand wins): written as asm because the compiler otherwise puts a
// canonicalising `v_max_f32 x, x, x` in front of every operand it cannot prove quiet - 24 of them per wide record.
#if defined(__HIP_DEVICE_COMPILE__)
PT_HD float ptc_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PT_HD float ptc_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PT_HD float ptc_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
PT_HD float ptc_max3_0(float a, float b) { float r; asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
PT_HD float ptc_min(float a, float b) { return pt_fmin(a, b); }
PT_HD float ptc_max(float a, float b) { return pt_fmax(a, b); }
PT_HD float ptc_min3(float a, float b, float c) { return pt_fmin(pt_fmin(a, b), c); }
PT_HD float ptc_max3_0(float a, float b) { return pt_fmax(pt_fmax(a, b), 0.0f); }
#endif
// bvh.rs:394-405 on per-axis (near, far) pairs: tmin = max(max(max(nx, ny), nz), 0), tmax = min(min(fx, fy), fz)
PT_HD bool ptc_close(float nx, float fx, float ny, float fy, float nz, float fz, float& tmin_out) {
    const float tmin = ptc_max3_0(ptc_max(nx, ny), nz);
    const float tmax = ptc_min3(fx, fy, fz);
    tmin_out = tmin;
    return tmax >= tmin;
}

// The decision stage: per-axis (near, far) of the four slots in, stack entries and next reference out.
struct PtcSlots {
    float nx0, fx0, nx1, fx1, nx2, fx2, nx3, fx3;
    float ny0, fy0, ny1, fy1, ny2, fy2, ny3, fy3;
    float nz0, fz0, nz1, fz1, nz2, fz2, nz3, fz3;
};

// The decisions of a record are boolean algebra over ten comparisons.  On the device a per-lane boolean IS a lane mask in a
// scalar register pair, but LLVM canonicalises (c && a) || (!c && b) into a select of i1 and the backend lowers that through
// 0 / 1 VGPRs (five VALU instructions per select, twenty per record).  So the device build does the algebra on the masks
// themselves - ballot of a comparison is the comparison's own result register, and / or / andn2 are scalar instructions,
// inverse ballot hands a mask back as a boolean - and the host build (the model of the CPU tests) on plain bools.
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long PtcMask;
#define PTC_B(x) __builtin_amdgcn_ballot_w64(x)
#define PTC_NOT(m) (~(m))
#define PTC_IB(m) __builtin_amdgcn_inverse_ballot_w64(m)
#else
typedef bool PtcMask;
#define PTC_B(x) (x)
#define PTC_NOT(m) (!(m))
#define PTC_IB(m) (m)
#endif

template <class F4>
PT_HD void ptc_wide_decide(const PtcSlots& q, F4 rw, PtcWideStep& s) {
    float k0, k1, k2, k3, kR, kL;
    const PtcMask h0 = PTC_B(ptc_close(q.nx0, q.fx0, q.ny0, q.fy0, q.nz0, q.fz0, k0));
    const PtcMask h1 = PTC_B(ptc_close(q.nx1, q.fx1, q.ny1, q.fy1, q.nz1, q.fz1, k1));
    const PtcMask h2 = PTC_B(ptc_close(q.nx2, q.fx2, q.ny2, q.fy2, q.nz2, q.fz2, k2));
    const PtcMask h3 = PTC_B(ptc_close(q.nx3, q.fx3, q.ny3, q.fy3, q.nz3, q.fz3, k3));
    // the children's own boxes (joins of their slots), bvh.rs:260-261 at X
    const bool bR = ptc_close(ptc_min(q.nx0, q.nx1), ptc_max(q.fx0, q.fx1), ptc_min(q.ny0, q.ny1), ptc_max(q.fy0, q.fy1),
                              ptc_min(q.nz0, q.nz1), ptc_max(q.fz0, q.fz1), kR);
    const bool bL = ptc_close(ptc_min(q.nx2, q.nx3), ptc_max(q.fx2, q.fx3), ptc_min(q.ny2, q.ny3), ptc_max(q.fy2, q.fy3),
                              ptc_min(q.nz2, q.nz3), ptc_max(q.fz2, q.fz3), kL);
    const PtcMask hR = PTC_B(bR), hL = PTC_B(bL);
    const uint32_t r0 = pt_f2u(rw.x), r1 = pt_f2u(rw.y), r2 = pt_f2u(rw.z), r3 = pt_f2u(rw.w);
    // inside each child: left = the odd slot, right = the even one; left first iff l_min < r_min (bvh.rs:266)
    const PtcMask mRl = h1 & (PTC_B(k1 < k0) | PTC_NOT(h0)), mLl = h3 & (PTC_B(k3 < k2) | PTC_NOT(h2));
    const bool Rl = PTC_IB(mRl), Ll = PTC_IB(mLl);
    const uint32_t Rnear = Rl ? r1 : r0, Rfar = Rl ? r0 : r1, Lnear = Ll ? r3 : r2, Lfar = Ll ? r2 : r3;
    const float Rfar_e = Rl ? k0 : k1, Lfar_e = Ll ? k2 : k3;
    const PtcMask Rboth = h0 & h1, Lboth = h2 & h3, Rany = h0 | h1, Lany = h2 | h3;
    // at X
    const PtcMask both = hL & hR, any = hL | hR;
    const PtcMask mNearL = hL & (PTC_B(kL < kR) | PTC_NOT(hR)), mNearR = PTC_NOT(mNearL);
    const PtcMask Nany = (mNearL & Lany) | (mNearR & Rany), Nboth = (mNearL & Lboth) | (mNearR & Rboth);
    const PtcMask Fany = (mNearL & Rany) | (mNearR & Lany), mFboth = (mNearL & Rboth) | (mNearR & Lboth);
    const bool nearL = PTC_IB(mNearL), Fboth = PTC_IB(mFboth);
    s.v0 = PTC_IB(both & mFboth);
    s.ref0 = nearL ? Rfar : Lfar;
    s.e0 = nearL ? Rfar_e : Lfar_e;
    s.v1 = PTC_IB(both & Fany);
    const uint32_t fnear = nearL ? Rnear : Lnear;
    s.ref1 = Fboth ? (fnear | PTC_COMPANION) : fnear;
    s.e1 = nearL ? kR : kL;
    s.v2 = PTC_IB(any & Nboth);
    s.ref2 = nearL ? Lfar : Rfar;
    s.e2 = nearL ? Lfar_e : Rfar_e;
    const uint32_t nnear = nearL ? Lnear : Rnear;
    s.cur = PTC_IB(any & Nany) ? nnear : PTC_IDLE;
    s.hit_l = bL; s.hit_r = bR; s.entry_l = kL; s.entry_r = kR;
}

// General form: any REGULAR ray.  (near, far) per axis = (minNum, maxNum) of the two plane distances, as bvh.rs:394-401.
template <class F4>
PT_HD void ptc_wide_step(const DRay& r, F4 lox, F4 loy, F4 loz, F4 hix, F4 hiy, F4 hiz, F4 rw, PtcWideStep& s) {
    const PtcPair ax = ptc_slab_axis2(lox.x, lox.y, hix.x, hix.y, r.ox, r.ix), bx = ptc_slab_axis2(lox.z, lox.w, hix.z, hix.w, r.ox, r.ix);
    const PtcPair ay = ptc_slab_axis2(loy.x, loy.y, hiy.x, hiy.y, r.oy, r.iy), by = ptc_slab_axis2(loy.z, loy.w, hiy.z, hiy.w, r.oy, r.iy);
    const PtcPair az = ptc_slab_axis2(loz.x, loz.y, hiz.x, hiz.y, r.oz, r.iz), bz = ptc_slab_axis2(loz.z, loz.w, hiz.z, hiz.w, r.oz, r.iz);
    PtcSlots q;
    q.nx0 = ptc_min(ax.t1.x, ax.t2.x); q.fx0 = ptc_max(ax.t1.x, ax.t2.x); q.nx1 = ptc_min(ax.t1.y, ax.t2.y); q.fx1 = ptc_max(ax.t1.y, ax.t2.y);
    q.nx2 = ptc_min(bx.t1.x, bx.t2.x); q.fx2 = ptc_max(bx.t1.x, bx.t2.x); q.nx3 = ptc_min(bx.t1.y, bx.t2.y); q.fx3 = ptc_max(bx.t1.y, bx.t2.y);
    q.ny0 = ptc_min(ay.t1.x, ay.t2.x); q.fy0 = ptc_max(ay.t1.x, ay.t2.x); q.ny1 = ptc_min(ay.t1.y, ay.t2.y); q.fy1 = ptc_max(ay.t1.y, ay.t2.y);
    q.ny2 = ptc_min(by.t1.x, by.t2.x); q.fy2 = ptc_max(by.t1.x, by.t2.x); q.ny3 = ptc_min(by.t1.y, by.t2.y); q.fy3 = ptc_max(by.t1.y, by.t2.y);
    q.nz0 = ptc_min(az.t1.x, az.t2.x); q.fz0 = ptc_max(az.t1.x, az.t2.x); q.nz1 = ptc_min(az.t1.y, az.t2.y); q.fz1 = ptc_max(az.t1.y, az.t2.y);
    q.nz2 = ptc_min(bz.t1.x, bz.t2.x); q.fz2 = ptc_max(bz.t1.x, bz.t2.x); q.nz3 = ptc_min(bz.t1.y, bz.t2.y); q.fz3 = ptc_max(bz.t1.y, bz.t2.y);
    ptc_wide_decide(q, rw, s);
}

// Sign-select form, for a ray whose inverse direction is finite and non-zero on every axis (PTC_RAY_SIGN_SELECT) over boxes
// with min <= max (every box BVH::build makes; pt_scene_create checks a caller's): t = fl(fl(c - o) * inv) is then monotone
// in c and never NaN for a finite c, so min(t(lo), t(hi)) IS t(lo) where inv > 0 and t(hi) where inv < 0 - the caller loads
// the near planes (lo or hi by the sign of inv, per axis) and the far planes and the 24 min / max of the general form drop
// out.  An empty slot's NaN planes give NaN either way.  Same values bit for bit, so the same decisions;
// tests/cpp/scoped_model.cpp runs both forms on every record such a ray visits and compares.
template <class F4>
PT_HD void ptc_wide_step_sign(const DRay& r, F4 nearx, F4 neary, F4 nearz, F4 farx, F4 fary, F4 farz, F4 rw, PtcWideStep& s) {
    const PtcPair ax = ptc_slab_axis2(nearx.x, nearx.y, farx.x, farx.y, r.ox, r.ix), bx = ptc_slab_axis2(nearx.z, nearx.w, farx.z, farx.w, r.ox, r.ix);
    const PtcPair ay = ptc_slab_axis2(neary.x, neary.y, fary.x, fary.y, r.oy, r.iy), by = ptc_slab_axis2(neary.z, neary.w, fary.z, fary.w, r.oy, r.iy);
    const PtcPair az = ptc_slab_axis2(nearz.x, nearz.y, farz.x, farz.y, r.oz, r.iz), bz = ptc_slab_axis2(nearz.z, nearz.w, farz.z, farz.w, r.oz, r.iz);
    PtcSlots q;
    q.nx0 = ax.t1.x; q.fx0 = ax.t2.x; q.nx1 = ax.t1.y; q.fx1 = ax.t2.y; q.nx2 = bx.t1.x; q.fx2 = bx.t2.x; q.nx3 = bx.t1.y; q.fx3 = bx.t2.y;
    q.ny0 = ay.t1.x; q.fy0 = ay.t2.x; q.ny1 = ay.t1.y; q.fy1 = ay.t2.y; q.ny2 = by.t1.x; q.fy2 = by.t2.x; q.ny3 = by.t1.y; q.fy3 = by.t2.y;
    q.nz0 = az.t1.x; q.fz0 = az.t2.x; q.nz1 = az.t1.y; q.fz1 = az.t2.y; q.nz2 = bz.t1.x; q.fz2 = bz.t2.x; q.nz3 = bz.t1.y; q.fz3 = bz.t2.y;
    ptc_wide_decide(q, rw, s);
}
// the ray qualifies for the sign-select form: every inverse direction component finite and non-zero
PT_HD bool ptc_ray_sign_select(const DRay& r) {
    const float big = pt_inf();
    return pt_fabs(r.ix) < big && pt_fabs(r.iy) < big && pt_fabs(r.iz) < big && r.ix != 0.0f && r.iy != 0.0f && r.iz != 0.0f;
}

#endif
