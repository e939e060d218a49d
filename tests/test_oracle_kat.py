"""Pins the CPU oracle against every known-answer vector the reference holds for the hot path.

  triangle.rs:130-216   Triangle::new + intersect: one exact hit (t=1, u=v=1/3) and four misses
  camera.rs:47-66       corner ray directions — up to scale: the reference's test normalises, its
                        generate_ray does not, so direction ratios are compared
  color.rs:184-218      sRGB -> XYZ primaries (chromaticity / luminance, abs eps 1e-3)

plus the oracle's own cross-checks (recursive traversal == brute force, slab-test NaN rules).
No GPU involved.
"""
import numpy as np
import pytest

from pathtracer_amd import abi, core, workloads

FRONT = ((0, 0, -1), (1, 0, -1), (0, 1, -1))
THIRD = np.float32(1.0) / np.float32(3.0)
TWO3 = np.float32(2.0) / np.float32(3.0)


# ---- triangle.rs:130-216 ---------------------------------------------------------------------

def test_triangle_intersection_kat(oracle):
    # triangle.rs:130-148 — assert_eq!, i.e. exact
    hit = oracle.triangle_intersect(*FRONT, (THIRD, THIRD, 0), (0, 0, -1))
    assert hit is not None
    assert hit[0] == np.float32(1.0) and hit[1] == THIRD and hit[2] == THIRD


def test_triangle_miss_t_negative(oracle):
    # triangle.rs:150-165
    assert oracle.triangle_intersect((0, 0, 2), (1, 0, 2), (0, 1, 2), (TWO3, TWO3, 0), (0, 0, 1)) is None


def test_triangle_miss_u_plus_v_greater_than_one(oracle):
    # triangle.rs:167-182
    assert oracle.triangle_intersect(*FRONT, (TWO3, TWO3, 0), (0, 0, 1)) is None


def test_triangle_miss_u_negative(oracle):
    # triangle.rs:184-199
    assert oracle.triangle_intersect(*FRONT, (-TWO3, TWO3, 0), (0, 0, 1)) is None


def test_triangle_miss_v_negative(oracle):
    # triangle.rs:201-216
    assert oracle.triangle_intersect(*FRONT, (TWO3, -TWO3, 0), (0, 0, 1)) is None


def test_triangle_self_hit_epsilon(oracle):
    """relative_eq!(distance, 0.0) (triangle.rs:110, approx 0.5.1 defaults): |t| <= f32::EPSILON is a miss."""
    eps = np.float32(1.1920929e-7)
    assert oracle.triangle_intersect(*FRONT, (0.25, 0.25, -1 + float(eps)), (0, 0, -1)) is None
    assert oracle.triangle_intersect(*FRONT, (0.25, 0.25, -1 + 4 * float(eps)), (0, 0, -1)) is not None


def test_triangle_precompute_matches_definition(oracle):
    """triangle.rs:50-60 against float64: n = ab x ac, n1 = ac x n / |n|^2, n2 = n x ab / |n|^2."""
    rng = np.random.default_rng(0)
    for _ in range(50):
        t = np.zeros((), dtype=abi.TRIANGLE)
        pts = rng.uniform(-2, 2, size=(3, 3)).astype(np.float32)
        t["a"], t["b"], t["c"] = pts
        got = oracle.triangle_precompute(t).astype(np.float64)
        a, b, c = pts.astype(np.float64)
        n = np.cross(b - a, c - a)
        n1 = np.cross(c - a, n) / n.dot(n)
        n2 = np.cross(n, b - a) / n.dot(n)
        want = np.concatenate([n, [a.dot(n)], n1, [-a.dot(n1)], n2, [-a.dot(n2)]])
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5)


# ---- camera.rs:47-66 --------------------------------------------------------------------------

@pytest.mark.parametrize("xy,want", [
    ((-0.5, -0.5), (-18.0, 13.5, -50.0)), ((1024.5, -0.5), (18.0, 13.5, -50.0)),
    ((-0.5, 768.5), (-18.0, -13.5, -50.0)), ((1024.5, 768.5), (18.0, -13.5, -50.0))])
def test_camera_corners(oracle, xy, want):
    d = oracle.camera_ray(1024, 768, 36.0, 50.0, *xy).astype(np.float64)
    w = np.array(want)
    np.testing.assert_allclose(d / np.linalg.norm(d), w / np.linalg.norm(w), atol=1e-6)
    assert d[2] == -50.0          # generate_ray does NOT normalise (camera.rs:19-28)


def test_camera_rays_of_the_library_match_the_oracle(oracle):
    """pt_gen_primary_rays (host helper of the product) == Camera::generate_ray(x, y) at pixel centres."""
    rays = core.gen_primary_rays(37, 23, 36.0, 50.0)
    assert (rays["o"] == 0).all()
    for (x, y) in [(0, 0), (36, 0), (0, 22), (36, 22), (17, 11)]:
        assert np.array_equal(rays["d"][y * 37 + x], oracle.camera_ray(37, 23, 36.0, 50.0, x, y))


# ---- color.rs:184-218 --------------------------------------------------------------------------

@pytest.mark.parametrize("rgb,x,y,Y", [
    ((1, 0, 0), 0.64, 0.33, 0.2126), ((0, 1, 0), 0.30, 0.60, 0.7152),
    ((0, 0, 1), 0.15, 0.06, 0.0722), ((1, 1, 1), 0.3127, 0.3290, 1.0)])
def test_srgb_primaries(oracle, rgb, x, y, Y):
    X, Yv, Z = oracle.srgb_to_xyz(rgb).astype(np.float64)
    s = X + Yv + Z
    assert abs(X / s - x) < 1e-3 and abs(Yv / s - y) < 1e-3 and abs(Yv - Y) < 1e-3


# ---- slab test (bvh.rs:390-410) ------------------------------------------------------------------

def test_slab_basic_and_nan_rules(oracle):
    assert oracle.ray_aabb((0, 0, 0), (0, 0, -1), (-1, -1, -3), (1, 1, -2)) == np.float32(2.0)
    assert oracle.ray_aabb((0, 0, 0), (0, 0, 1), (-1, -1, -3), (1, 1, -2)) is None
    # origin inside: tmin clamps to 0 (bvh.rs:403)
    assert oracle.ray_aabb((0, 0, -2.5), (0, 0, -1), (-1, -1, -3), (1, 1, -2)) == np.float32(0.0)
    # d.x == 0 strictly inside the slab: t1 = -inf, t2 = +inf on that axis, the other axes decide
    assert oracle.ray_aabb((0.5, 0, 0), (0, 0, -1), (-1, -1, -3), (1, 1, -2)) == np.float32(2.0)
    # d.x == 0 with the origin ON the min plane: t1 = 0*inf = NaN, t2 = +inf; f32::min/max return
    # the non-NaN operand, so tmin = tmax = +inf on x and the finite exit of y/z is below it: miss.
    # On the max plane t1 = -inf, t2 = NaN -> tmax = -inf: miss as well.  (By hand from bvh.rs:391-409.)
    assert oracle.ray_aabb((-1, 0, 0), (0, 0, -1), (-1, -1, -3), (1, 1, -2)) is None
    assert oracle.ray_aabb((1, 0, 0), (0, 0, -1), (-1, -1, -3), (1, 1, -2)) is None
    # d.x == 0 outside the slab: +-inf on that axis -> miss
    assert oracle.ray_aabb((-2, 0, 0), (0, 0, -1), (-1, -1, -3), (1, 1, -2)) is None
    # un-normalised direction: the entry distance is the ray PARAMETER
    assert oracle.ray_aabb((0, 0, 0), (0, 0, -50), (-1, -1, -3), (1, 1, -2)) == np.float32(2.0) / np.float32(50.0)


# ---- traversal == brute force ---------------------------------------------------------------------

def _bounce_rays(rng, n):
    rays = np.zeros(n, dtype=abi.RAY)
    rays["o"] = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32) + np.float32([0, 0, -3])
    rays["d"] = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    return rays


@pytest.mark.parametrize("n_tris", [1, 2, 3, 17, 2000])
def test_recursive_traversal_equals_brute_force(oracle, n_tris):
    arr = workloads.soup_scene(n_tris)
    sc = oracle.scene(arr)
    rays = np.concatenate([core.gen_primary_rays(64, 36), _bounce_rays(np.random.default_rng(n_tris), 2000)])
    a, b = oracle.trace(sc, rays), oracle.trace_brute(sc, rays)
    # equal minimum distance everywhere; ids may only differ on exact ties (none in random soup)
    assert np.array_equal(a["t"], b["t"])
    assert np.array_equal(a["tri"], b["tri"])
    if n_tris >= 2000:
        assert (a["tri"] != abi.PT_NO_TRIANGLE).mean() > 0.05


def test_start_triangle_is_excluded_by_identity(oracle):
    arr = workloads.soup_scene(500)
    sc = oracle.scene(arr)
    rays = core.gen_primary_rays(96, 54)
    first = oracle.trace(sc, rays)
    hit = first["tri"] != abi.PT_NO_TRIANGLE
    again = oracle.trace(sc, rays[hit], first["tri"][hit])
    assert (again["tri"] != first["tri"][hit]).all()
    assert np.array_equal(again, oracle.trace_brute(sc, rays[hit], first["tri"][hit]))
    # whatever is hit instead lies behind the excluded triangle
    still = again["tri"] != abi.PT_NO_TRIANGLE
    assert (again["t"][still] >= first["t"][hit][still]).all()


def test_global_best_traversal_returns_the_same_hits(oracle):
    """The FAST device traversal's rule (one global best distance), run on the CPU, agrees with the
    reference recursion on ids and t, and visits no more nodes."""
    arr = workloads.soup_scene(5000)
    sc = oracle.scene(arr)
    rays = core.gen_primary_rays(128, 72)
    ref, cnt = oracle.trace(sc, rays, count=True)
    gb, gcnt = oracle.trace_globalbest_stats(sc, rays)
    assert np.array_equal(ref, gb)
    assert gcnt[0] <= cnt[0] and gcnt[1] <= cnt[1]


def test_empty_and_single_triangle_scenes(oracle):
    rays = core.gen_primary_rays(16, 16)
    sc = oracle.scene(workloads.empty_scene())
    assert sc.nodes().size == 0                                         # bvh.rs:353-355
    assert (oracle.trace(sc, rays)["tri"] == abi.PT_NO_TRIANGLE).all()    # bvh.rs:364-368
    one = oracle.scene(workloads.soup_scene(1))
    nodes = one.nodes()
    assert nodes.size == 1 and nodes[0]["is_leaf"] == 1                  # bvh.rs:154-159


def test_texture_bilinear_against_a_numpy_restatement(oracle):
    """Texture::Bilinear (material.rs:101-123): the reference defines it, its loader never builds one, and it holds no vector
    for it (parity unpinned) - the oracle's restatement is checked here against an independent one written from the same
    lines in numpy binary32: x = rem_euclid(u, 1) * W, y = (1 - rem_euclid(v, 1)) * H, the four texels around (floor x,
    floor y) with rem_euclid wrap, p0 = p00 (1 - tx) + p10 tx, p1 = p01 (1 - tx) + p11 tx, p0 (1 - ty) + p1 ty."""
    from pathtracer_amd import abi, core, workloads
    f32 = np.float32
    rng = np.random.default_rng(8)
    img3, img1 = rng.uniform(0, 1, size=(5, 7, 3)).astype(f32), rng.uniform(0, 1, size=(3, 4)).astype(f32)      # (H, W, ch)
    arr = abi.SceneArrays(workloads.box_triangles(), core.bvh_build(workloads.box_triangles()), workloads.box_materials(),
                          core.sky_d65(1.0), [img3, img1])
    sc = oracle.scene(arr)

    def rem1(x):
        r = np.fmod(f32(x), f32(1.0))
        return f32(r + f32(1.0)) if r < 0 else f32(r)

    def bilinear(img, u, v):
        h, w = img.shape[0], img.shape[1]
        x, y = f32(rem1(u) * f32(w)), f32(f32(f32(1.0) - rem1(v)) * f32(h))
        x0, y0 = int(np.floor(x)), int(np.floor(y))
        tx, ty = f32(x - np.floor(x)), f32(y - np.floor(y))
        px = lambda i, j: np.atleast_1d(img[j % h, i % w]).astype(f32)
        p0 = (px(x0, y0) * f32(f32(1.0) - tx)).astype(f32) + (px(x0 + 1, y0) * tx).astype(f32)
        p1 = (px(x0, y0 + 1) * f32(f32(1.0) - tx)).astype(f32) + (px(x0 + 1, y0 + 1) * tx).astype(f32)
        return ((p0 * f32(f32(1.0) - ty)).astype(f32) + (p1 * ty).astype(f32)).astype(f32)

    uvs = [(0.0, 0.0), (0.5, 0.5), (0.999999, 0.000001), (1.0, 1.0), (-0.25, 2.75), (3.1, -7.9), (0.1428571, 0.2)]
    uvs += [tuple(p) for p in rng.uniform(-3, 3, size=(200, 2))]
    for u, v in uvs:
        got3 = oracle.texture_sample(sc, abi.PT_TEX_BILINEAR, 0, u, v)
        assert np.array_equal(got3.view(np.uint32), bilinear(img3, u, v).view(np.uint32)), (u, v)
        got1 = oracle.texture_sample(sc, abi.PT_TEX_BILINEAR, 1, u, v)
        assert got1[0].view(np.uint32) == bilinear(img1, u, v)[0].view(np.uint32) and got1[0] == got1[1] == got1[2], (u, v)
        # Nearest keeps its old meaning (material.rs:92-100)
        h, w = img3.shape[:2]
        x, y = int(np.floor(f32(rem1(u) * f32(w)))), int(np.floor(f32(f32(f32(1.0) - rem1(v)) * f32(h))))
        assert np.array_equal(oracle.texture_sample(sc, abi.PT_TEX_NEAREST, 0, u, v), img3[y % h, x % w]), (u, v)
    # at a texel corner the four weights collapse to one texel; halfway it is the mean of two
    assert np.allclose(oracle.texture_sample(sc, abi.PT_TEX_BILINEAR, 0, 2.5 / 7, 1 - 1.0 / 5), 0.5 * (img3[1, 2] + img3[1, 3]), atol=1e-6)
