"""Parity of the HIP closest-hit path (pt_trace_closest, through the C ABI) with the CPU oracle.

Bar (north_star): hit triangle ids bit-exact, t within 1e-5 relative.  Because the kernels repeat
the reference's float operations in the same order, t/u/v are in fact compared for exact equality.
"""
import numpy as np
import pytest

from pathtracer_amd import abi, core, workloads
from tests.golden import fixture_inputs as fi

pytestmark = pytest.mark.gpu

MODES = [abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY, abi.PT_TRAVERSE_MIN_T]


def bounce_rays(rng, hits, rays, n):
    """Secondary rays as the integrator makes them: origin on a hit point, un-normalised direction
    in [-2,2]^3-ish, start = the triangle just hit (scene.rs:355-356)."""
    ok = np.flatnonzero(hits["tri"] != abi.PT_NO_TRIANGLE)
    pick = rng.choice(ok, size=n)
    out = np.zeros(n, dtype=abi.RAY)
    out["o"] = rays["o"][pick] + rays["d"][pick] * hits["t"][pick, None]
    out["d"] = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32) * rng.uniform(0.01, 2.0, size=(n, 1)).astype(np.float32)
    return out, hits["tri"][pick].astype(np.uint32)


def assert_same_hits(got, want, what):
    bad = np.flatnonzero(got["tri"] != want["tri"])
    assert bad.size == 0, f"{what}: {bad.size} id mismatches, first at ray {bad[:5]}: got {got[bad[:5]]}, want {want[bad[:5]]}"
    hit = want["tri"] != abi.PT_NO_TRIANGLE
    rel = np.abs(got["t"][hit] - want["t"][hit]) / np.maximum(np.abs(want["t"][hit]), 1e-30)
    assert (rel <= 1e-5).all(), f"{what}: t off by {rel.max()}"
    # stronger than required: same operations in the same order -> same bits
    assert np.array_equal(got["t"][hit], want["t"][hit]), what
    assert np.array_equal(got["u"][hit], want["u"][hit]) and np.array_equal(got["v"][hit], want["v"][hit]), what


@pytest.fixture(scope="module")
def soup(oracle, gpu_ctx):
    arr = workloads.soup_scene(200_000)
    return arr, oracle.scene(arr), gpu_ctx.scene(arr)


@pytest.mark.parametrize("mode", MODES)
def test_primary_rays(oracle, gpu_ctx, soup, mode):
    arr, osc, gsc = soup
    rays = core.gen_primary_rays(640, 360)
    gpu_ctx.set_traversal(mode)
    got = gpu_ctx.trace(gsc, rays)
    want = oracle.trace(osc, rays)
    assert (want["tri"] != abi.PT_NO_TRIANGLE).mean() > 0.9
    assert_same_hits(got, want, f"primary mode={mode}")


@pytest.mark.parametrize("mode", MODES)
def test_bounce_rays_with_start(oracle, gpu_ctx, soup, mode):
    arr, osc, gsc = soup
    rng = np.random.default_rng(7)
    prim = core.gen_primary_rays(320, 180)
    first = oracle.trace(osc, prim)
    rays, start = bounce_rays(rng, first, prim, 100_000)
    gpu_ctx.set_traversal(mode)
    got = gpu_ctx.trace(gsc, rays, start)
    want = oracle.trace(osc, rays, start)
    assert_same_hits(got, want, f"bounce mode={mode}")
    # the start triangle is excluded by identity only (bvh.rs:329-335)
    assert (got["tri"] != start).all()


@pytest.mark.parametrize("mode", MODES)
def test_axis_aligned_and_degenerate_directions(oracle, gpu_ctx, soup, mode):
    """Zero direction components give +-inf inverse directions and 0*inf = NaN in the slab test."""
    arr, osc, gsc = soup
    rng = np.random.default_rng(3)
    n = 20_000
    rays = np.zeros(n, dtype=abi.RAY)
    rays["o"] = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32) + np.float32([0, 0, -3])
    d = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    kill = rng.integers(0, 3, size=n)
    d[np.arange(n), kill] = 0.0
    d[: n // 4, (kill[: n // 4] + 1) % 3] = 0.0        # two zero components
    d[:8] = 0.0                                          # all-zero direction
    d[8:16] = np.float32([-0.0, 0.0, -1.0])
    rays["d"] = d
    gpu_ctx.set_traversal(mode)
    assert_same_hits(gpu_ctx.trace(gsc, rays), oracle.trace(osc, rays), f"axis-aligned mode={mode}")


@pytest.mark.parametrize("mode", MODES)
def test_deep_stack_spills(oracle, gpu_ctx, mode):
    """Heavily overlapping large triangles push the per-lane stack past the LDS ring (12 entries in
    the persistent FAST kernel, 8 in the others), so entries spill to and come back from HBM."""
    tris = core.gen_soup(11, 50_000, (-1, -1, -4), (1, 1, -2), 2.0, 0.0, 8)
    arr = abi.SceneArrays(tris, core.bvh_build(tris), workloads.soup_materials(), core.sky_d65(0.0))
    osc, gsc = oracle.scene(arr), gpu_ctx.scene(arr)
    rays = core.gen_primary_rays(256, 256)
    _, cnt = oracle.trace_globalbest_stats(osc, rays[::16])
    assert cnt[3] > 12 and cnt[2] / rays[::16].size > 12, f"scene does not overflow the ring (stack max {cnt[3]})"
    gpu_ctx.set_traversal(mode)
    assert_same_hits(gpu_ctx.trace(gsc, rays), oracle.trace(osc, rays), f"deep mode={mode}")


@pytest.mark.parametrize("mode", MODES)
def test_reference_triangle_kats_on_device(gpu_ctx, mode):
    """triangle.rs:130-216 through the kernels: one exact hit, four misses."""
    def one(tri_pts, o, d):
        t = np.zeros(1, dtype=abi.TRIANGLE)
        t["a"], t["b"], t["c"] = tri_pts
        arr = abi.SceneArrays(t, core.bvh_build(t), np.array([abi.default_material()], dtype=abi.MATERIAL), core.sky_d65(0.0))
        sc = gpu_ctx.scene(arr)
        r = np.zeros(1, dtype=abi.RAY)
        r["o"], r["d"] = o, d
        return gpu_ctx.trace(sc, r)[0]
    gpu_ctx.set_traversal(mode)
    third = np.float32(1.0) / np.float32(3.0)
    front = ((0, 0, -1), (1, 0, -1), (0, 1, -1))
    h = one(front, (third, third, 0), (0, 0, -1))
    assert h["tri"] == 0 and h["t"] == np.float32(1.0) and h["u"] == third and h["v"] == third
    two3 = np.float32(2.0) / np.float32(3.0)
    assert one(((0, 0, 2), (1, 0, 2), (0, 1, 2)), (two3, two3, 0), (0, 0, 1))["tri"] == abi.PT_NO_TRIANGLE
    assert one(front, (two3, two3, 0), (0, 0, 1))["tri"] == abi.PT_NO_TRIANGLE
    assert one(front, (-two3, two3, 0), (0, 0, 1))["tri"] == abi.PT_NO_TRIANGLE
    assert one(front, (two3, -two3, 0), (0, 0, 1))["tri"] == abi.PT_NO_TRIANGLE


def test_empty_scene_and_empty_batch(gpu_ctx):
    sc = gpu_ctx.scene(workloads.empty_scene())
    rays = core.gen_primary_rays(64, 64)
    got = gpu_ctx.trace(sc, rays)
    assert (got["tri"] == abi.PT_NO_TRIANGLE).all()
    assert gpu_ctx.trace(sc, rays[:0]).size == 0


def test_bad_scene_is_an_error_code(gpu_ctx):
    arr = workloads.soup_scene(100)
    bad = abi.SceneArrays(arr.triangles, arr.nodes.copy(), arr.materials, arr.sky)
    bad.nodes["child_l_index"][0] = 0          # self-loop: traversal would never terminate
    with pytest.raises(core.PtError) as e:
        gpu_ctx.scene(bad)
    assert e.value.code == abi.PT_ERR_INVALID
    bad2 = abi.SceneArrays(arr.triangles, arr.nodes[:-1], arr.materials, arr.sky)
    with pytest.raises(core.PtError):
        gpu_ctx.scene(bad2)
    # a child referenced by two parents leaves another node (here a leaf) unreachable
    bad3 = abi.SceneArrays(arr.triangles, arr.nodes.copy(), arr.materials, arr.sky)
    internal = np.flatnonzero(bad3.nodes["is_leaf"] == 0)
    leaf_parent = next(i for i in internal[::-1] if bad3.nodes["is_leaf"][bad3.nodes["child_l_index"][i]]
                       and bad3.nodes["is_leaf"][bad3.nodes["child_r_index"][i]])
    bad3.nodes["child_l_index"][leaf_parent] = bad3.nodes["child_r_index"][leaf_parent]
    with pytest.raises(core.PtError) as e:
        gpu_ctx.scene(bad3)
    assert e.value.code == abi.PT_ERR_INVALID and "pt_scene_create" in str(e.value)


# ---- device-side BVH build (pt_scene_create_ex, PT_SCENE_BUILD_LBVH) --------------------------------------

def same_up_to_ties(got, want):
    """Another tree over the same triangles: the hit DISTANCE is the same to the bit on every ray; the triangle
    (hence u, v) may differ only where two triangles are hit at exactly that distance (soup-1M at 1080p: 1 ray of
    2 073 600, tools/lbvh_bench.py)."""
    assert got["t"].tobytes() == want["t"].tobytes()
    tie = got["tri"] != want["tri"]
    assert tie.sum() <= 2 and not (tie & ((got["tri"] == abi.PT_NO_TRIANGLE) | (want["tri"] == abi.PT_NO_TRIANGLE))).any()
    assert got[~tie].tobytes() == want[~tie].tobytes()


@pytest.mark.parametrize("n_tris", [2, 3, 5, 64, 20_000, 300_000])
def test_lbvh_scene_returns_the_same_hits(oracle, gpu_ctx, n_tris):
    """Another topology, same triangles: the minimum-t hit is the same, to the bit, in every traversal mode."""
    arr = workloads.soup_scene(n_tris)
    gsc = gpu_ctx.scene(arr, lbvh=True)
    assert gsc.info().n_internal_nodes == n_tris - 1 and gsc.info().max_depth >= int(np.ceil(np.log2(n_tris)))
    rays = core.gen_primary_rays(320, 180)
    want = oracle.trace(oracle.scene(arr), rays)
    for mode in MODES:
        gpu_ctx.set_traversal(mode)
        same_up_to_ties(gpu_ctx.trace(gsc, rays), want)
    gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
    hit = want["tri"] != abi.PT_NO_TRIANGLE
    if hit.any():
        brays, start = fi.bounce_rays(rays, want)
        same_up_to_ties(gpu_ctx.trace(gsc, brays, start), oracle.trace(oracle.scene(arr), brays, start))


def test_lbvh_duplicate_centres_and_degenerate_extent(oracle, gpu_ctx):
    """Every Morton key equal (all triangles share one box centre) and a zero-extent axis: the radix tree falls
    back to positions and still holds every triangle once."""
    arr = workloads.soup_scene(500)
    t = arr.triangles.copy()
    c = (np.minimum(np.minimum(t["a"], t["b"]), t["c"]) + np.maximum(np.maximum(t["a"], t["b"]), t["c"])) * np.float32(0.5)
    shift = np.array([0, 0, -3], dtype=np.float32) - c
    for k in "abc":
        t[k] += shift
    arr2 = abi.SceneArrays(t, core.bvh_build(t), arr.materials, arr.sky)
    rays = core.gen_primary_rays(128, 72)
    want = oracle.trace(oracle.scene(arr2), rays)
    got = gpu_ctx.trace(gpu_ctx.scene(arr2, lbvh=True), rays)
    assert got["t"].tobytes() == want["t"].tobytes()          # an equal-t tie may pick another triangle; the distance may not differ
    assert (got["tri"] != want["tri"]).mean() < 0.01 and (want["tri"] != abi.PT_NO_TRIANGLE).mean() > 0.02


def test_lbvh_render_matches_the_oracle(oracle, gpu_ctx):
    arr = workloads.soup_scene(20_000, sky=workloads.hw_sky())
    got = gpu_ctx.render(gpu_ctx.scene(arr, lbvh=True), 80, 45, 0, 4, 4, seed=1)
    want = oracle.render(oracle.scene(arr), 80, 45, 0, 4, 4, seed=1)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(want))


def test_lbvh_rejects_nodes(gpu_ctx):
    arr = workloads.soup_scene(100)
    d = arr.desc()
    h = core.C.c_void_p()
    rc = core.lib().pt_scene_create_ex(gpu_ctx.handle, core.C.byref(d), abi.PT_SCENE_BUILD_LBVH, core.C.byref(h))
    assert rc == abi.PT_ERR_INVALID


@pytest.mark.parametrize("mode", MODES)
def test_coincident_triangles_resolve_as_in_the_recursion(oracle, gpu_ctx, mode):
    """Exactly duplicated triangles (equal boxes, equal t): the reference's recursion enters equal-distance boxes
    right before left and lets the later hit win (bvh.rs:266-270, 299-303).  Every mode, the 4-wide one included,
    returns the same triangle of each duplicate cluster as the CPU restatement of that recursion."""
    rng = np.random.default_rng(11)
    base = workloads.soup_scene(3000).triangles
    t = np.concatenate([base, base[rng.integers(0, base.size, 3000)], base[:500], base[:500]])       # clusters of 2..4 copies
    t = t[rng.permutation(t.size)]
    arr = abi.SceneArrays(t, core.bvh_build(t), workloads.soup_scene(8).materials, core.sky_d65(1.0))
    rays = core.gen_primary_rays(256, 144)
    want = oracle.trace(oracle.scene(arr), rays)
    assert (want["tri"] != abi.PT_NO_TRIANGLE).mean() > 0.5
    gpu_ctx.set_traversal(mode)
    got = gpu_ctx.trace(gpu_ctx.scene(arr), rays)
    gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
    assert got.tobytes() == want.tobytes()


def test_device_wide_collapse_is_byte_identical_to_the_host_loop(monkeypatch):
    """PTC_WIDE_COLLAPSE=2 builds the 4-wide records both ways (device kernels of lbvh.hip, host loop of core.hip) and fails
    scene creation unless the two arrays are byte-identical and the depths agree."""
    monkeypatch.setenv("PTC_WIDE_COLLAPSE", "2")
    ctx = core.Context(0)
    try:
        rng = np.random.default_rng(3)
        for n in (2, 3, 4, 5, 6, 7, 8, 33, 1000, 20_000, 300_000):
            arr = workloads.soup_scene(n)
            sc = ctx.scene(arr)
            assert sc.info().n_internal_nodes == n - 1
            sc.close()
        base = workloads.soup_scene(2000).triangles
        t = np.concatenate([base, base[rng.integers(0, base.size, 2000)]])         # duplicate clusters
        ctx.scene(abi.SceneArrays(t, core.bvh_build(t), workloads.soup_scene(8).materials, core.sky_d65(1.0))).close()
        ctx.scene(workloads.box_scene()).close()
    finally:
        ctx.close()


def chain_scene(n):
    """n triangles hung on a right-leaning chain: node 2k = Node{left: leaf 2k+1, right: 2k+2}; depth n - 1."""
    tris = workloads.soup_scene(n).triangles
    nodes = np.zeros(2 * n - 1, dtype=abi.BVH_NODE)
    lo = np.minimum(np.minimum(tris["a"], tris["b"]), tris["c"])
    hi = np.maximum(np.maximum(tris["a"], tris["b"]), tris["c"])
    suf_lo = np.minimum.accumulate(lo[::-1], axis=0)[::-1]          # box of triangles k..n-1
    suf_hi = np.maximum.accumulate(hi[::-1], axis=0)[::-1]
    for k in range(n - 1):
        nd = nodes[2 * k]
        nd["child_l_index"], nd["child_r_index"] = 2 * k + 1, 2 * k + 2
        nd["child_l_aabb"]["min"], nd["child_l_aabb"]["max"] = lo[k], hi[k]
        nd["child_r_aabb"]["min"], nd["child_r_aabb"]["max"] = suf_lo[k + 1], suf_hi[k + 1]
        nodes[2 * k + 1]["is_leaf"], nodes[2 * k + 1]["shape_index"] = 1, k
    nodes[2 * n - 2]["is_leaf"], nodes[2 * n - 2]["shape_index"] = 1, n - 1
    return abi.SceneArrays(tris, nodes, workloads.soup_scene(8).materials, core.sky_d65(1.0))


@pytest.mark.parametrize("flatten", [0, 1, 2])
def test_caller_nodes_are_validated_and_flattened_alike_on_device_and_host(monkeypatch, oracle, flatten):
    """PTC_FLATTEN: 0 = the kernels of lbvh.hip (default), 1 = the host loop of core.hip, 2 = both, and scene creation fails
    unless the two node arrays are byte-identical.  The verdicts on bad node vectors are the same either way."""
    monkeypatch.setenv("PTC_FLATTEN", str(flatten))
    ctx = core.Context(0)
    try:
        for n in (1, 2, 3, 5, 8, 33, 1000, 20_000, 300_000):
            arr = workloads.soup_scene(n)
            sc = ctx.scene(arr)
            assert sc.info().n_internal_nodes == n - 1
            if n == 20_000:
                rays = core.gen_primary_rays(96, 54)
                assert ctx.trace(sc, rays).tobytes() == oracle.trace(oracle.scene(arr), rays).tobytes()
                assert np.array_equal(sc.bvh(), arr.nodes)
            sc.close()
        ctx.scene(workloads.box_scene()).close()
        deep = chain_scene(600)                                   # depth 599: deep but legal
        sc = ctx.scene(deep)
        rays = core.gen_primary_rays(64, 36)
        assert sc.info().max_depth == 599 and ctx.trace(sc, rays).tobytes() == oracle.trace(oracle.scene(deep), rays).tobytes()
        sc.close()
        with pytest.raises(core.PtError) as e:                    # deeper than the traversal stacks are sized for
            ctx.scene(chain_scene(4200))
        assert e.value.code == abi.PT_ERR_UNSUPPORTED and "4096" in str(e.value)

        arr = workloads.soup_scene(500)

        def verdict(mutate):
            a = abi.SceneArrays(arr.triangles.copy(), arr.nodes.copy(), arr.materials.copy(), arr.sky)
            mutate(a.nodes)
            with pytest.raises(core.PtError) as e:
                ctx.scene(a)
            assert e.value.code == abi.PT_ERR_INVALID and "pt_scene_create" in str(e.value)
            return str(e.value)
        leaves, inner = np.flatnonzero(arr.nodes["is_leaf"]), np.flatnonzero(arr.nodes["is_leaf"] == 0)
        assert "shape_index" in verdict(lambda nd: nd["shape_index"].__setitem__(leaves[7], 500))
        assert "forward" in verdict(lambda nd: nd["child_l_index"].__setitem__(inner[5], inner[5]))          # a node that is its own child
        assert "forward" in verdict(lambda nd: nd["child_r_index"].__setitem__(inner[9], 999))               # past the end
        assert "forward" in verdict(lambda nd: nd["child_r_index"].__setitem__(inner[40], 0))                # back to the root
        assert "two leaves" in verdict(lambda nd: nd["shape_index"].__setitem__(leaves[3], nd["shape_index"][leaves[4]]))
        # two parents for one node: its sibling subtree is unreachable
        msg = verdict(lambda nd: nd["child_l_index"].__setitem__(inner[2], nd["child_r_index"][inner[2]]))
        assert "unreachable" in msg or "two leaves" in msg
        assert "internal nodes" in verdict(lambda nd: nd["is_leaf"].__setitem__(inner[-1], 1))                # a Node turned Leaf
    finally:
        ctx.close()


@pytest.mark.parametrize("n_tris", [0, 1, 2, 7, 20_000])
def test_scene_bvh_round_trip_of_caller_nodes(gpu_ctx, n_tris):
    """pt_scene_get_bvh undoes the flattening of pt_scene_create: the caller's `BVH.nodes` come back (Node fields of leaf
    entries zeroed, as pt_bvh_build leaves them)."""
    arr = workloads.soup_scene(n_tris) if n_tris else workloads.empty_scene()
    back = gpu_ctx.scene(arr).bvh()
    want = arr.nodes if arr.nodes is not None else np.zeros(0, dtype=abi.BVH_NODE)
    assert back.size == max(2 * n_tris - 1, 0) and back.tobytes() == want.tobytes()


@pytest.mark.parametrize("n_tris", [1, 2, 3, 64, 20_000, 300_000])
def test_device_built_tree_is_a_reference_tree(oracle, gpu_ctx, n_tris):
    """The device-built tree, read back as `BVH.nodes`, is a tree the reference's own recursion can walk: the CPU
    restatement of traverse_recursive over it returns what the GPU returns, bit for bit - ids at ties included in EXACT and
    REPLAY mode, and in MIN_T as well on these scenes - and creating a scene from the read-back nodes gives the same tree
    again."""
    arr = workloads.soup_scene(n_tris)
    gsc = gpu_ctx.scene(arr, lbvh=True)
    nodes = gsc.bvh()
    assert nodes.size == 2 * n_tris - 1 and int(nodes["is_leaf"].sum()) == n_tris
    assert np.array_equal(np.sort(nodes["shape_index"][nodes["is_leaf"] == 1]), np.arange(n_tris))
    arr2 = abi.SceneArrays(arr.triangles, nodes, arr.materials, arr.sky, arr.images)
    osc = oracle.scene(arr2)
    rays = core.gen_primary_rays(320, 180)
    want = oracle.trace(osc, rays)
    start = np.where(np.arange(rays.size) % 3 == 0, want["tri"], abi.PT_NO_TRIANGLE).astype(np.uint32)
    want_s = oracle.trace(osc, rays, start)
    try:
        for mode in MODES:
            gpu_ctx.set_traversal(mode)
            assert gpu_ctx.trace(gsc, rays).tobytes() == want.tobytes(), mode
            assert gpu_ctx.trace(gsc, rays, start).tobytes() == want_s.tobytes(), mode
    finally:
        gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
    assert gpu_ctx.scene(arr2).bvh().tobytes() == nodes.tobytes()


def test_the_sweeps_instructive_rays(oracle, gpu_ctx):
    """Rays the randomised sweep of round 1 found, kept as they were found.  EXACT (the default) and REPLAY
    return the recursion's answer on all of them, as on every other ray; MIN_T returns a hit of the same distance to
    within rounding.
    Seed 910, ray 18391: a flat axis-aligned triangle whose box is entered (by rounding) exactly at the t of a hit in ANOTHER
    subtree, 2 ulp after the triangle's own t: a kernel that prunes `best > entry` against the best hit of the whole ray
    loses it; the recursion, which only ever compares siblings, does not.
    Seed 1912, ray 14942: two coplanar triangles hit at the same t under different children of one binary node: entering the
    four grandchildren by their own entry distances keeps the other one of the tie.
    Seed 8824, rays 2669 and 11174: equal t behind an equal entry distance, and the same with a rounding gap on top."""
    from tests import fuzz_scenes
    for seed, special in ((910, [18391]), (1912, [14942]), (8824, [2669, 11174])):
        rng = np.random.default_rng(seed)
        tris, mats, sky, scale, images = fuzz_scenes.random_scene(rng)
        arr = abi.SceneArrays(tris, core.bvh_build(tris), mats, sky, images)
        rays = fuzz_scenes.random_rays(rng, 20000, scale)
        want = oracle.trace(oracle.scene(arr), rays)
        gsc = gpu_ctx.scene(arr)
        try:
            for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY):
                gpu_ctx.set_traversal(mode)
                assert gpu_ctx.trace(gsc, rays).tobytes() == want.tobytes(), (seed, mode)
            gpu_ctx.set_traversal(abi.PT_TRAVERSE_MIN_T)
            hits = gpu_ctx.trace(gsc, rays)
            others = ~np.isin(np.arange(rays.size), special)
            assert hits[others].tobytes() == want[others].tobytes(), seed
            for ray in special:
                assert (hits[ray]["tri"] == abi.PT_NO_TRIANGLE) == (want[ray]["tri"] == abi.PT_NO_TRIANGLE)
                assert abs(int(hits[ray:ray + 1]["t"].view(np.int32)[0]) - int(want[ray:ray + 1]["t"].view(np.int32)[0])) <= 4
        finally:
            gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)


@pytest.mark.parametrize("gen_seed", [5, 6, 7])
def test_irregular_rays_and_non_finite_hits_take_the_replay_path(oracle, gpu_ctx, gen_seed):
    """What EXACT hands to REPLAY inside the same call: rays with an infinite or NaN component, and rays that meet a
    triangle at a non-finite distance (a denormal determinant: 1 / det overflows, t * inf is inf or NaN) - where Some(inf),
    Some(NaN) and None part ways in bvh.rs:280-303.  Every mode that claims the recursion's answer must give it.

    The NaN-sign class: Triangle::intersect takes f32::is_sign_positive of values that can be NaN (triangle.rs:91-102: a NaN
    ray component, n1 / n2 = inf or NaN where |n|^2 underflows, inf - inf).  The sign bit of a NaN that an operation made
    or passed on is unspecified (IEEE 754; Rust's own docs call the result non-portable): x86 SSE makes negative default
    NaNs, gfx950 positive ones, so on such rays the reference's verdict depends on the CPU it runs on.  The oracle marks them
    (pto_trace_nan_flags); they are left out of the comparison, counted, and must still come back as a valid answer."""
    from tests import fuzz_scenes
    arr2, tiny, rays = fuzz_scenes.underflow_scene_and_rays(np.random.default_rng(gen_seed))
    n, t = rays.size, arr2.triangles
    want, nan_sign = oracle.trace_nan_flags(oracle.scene(arr2), rays)
    assert (~np.isfinite(want["t"]) & ~nan_sign).sum() > 100 and nan_sign.sum() > 5000        # Some(inf) results that ARE portable; the class that is not
    gsc = gpu_ctx.scene(arr2)
    try:
        for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY):
            gpu_ctx.set_traversal(mode)
            got = gpu_ctx.trace(gsc, rays)
            # NaN payloads of t / u / v may differ between x86 and gfx950: compare NaN as NaN
            same = (got["tri"] == want["tri"]) & np.all([(np.isnan(got[k]) & np.isnan(want[k])) | (got[k].view(np.uint32) == want[k].view(np.uint32)) for k in "tuv"], axis=0)
            bad = np.flatnonzero(~same & ~nan_sign)
            assert bad.size == 0, (mode, bad[:5], got[bad[:5]], want[bad[:5]], rays[bad[:5]])
            assert ((got["tri"] == abi.PT_NO_TRIANGLE) | (got["tri"] < t.size)).all()
            print(f"mode {mode}: {int(nan_sign.sum())} of {n} rays took the sign of a NaN; {int((~same & nan_sign).sum())} of those differ from the x86 oracle")
    finally:
        gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)


def loose_tree_scene(rng):
    """Duplicate clusters (equal-t ties: the visiting ORDER decides the answer) under a reference-built tree whose stored child
    boxes were then padded at random - still a valid BVH (every box contains its subtree), no longer the exact joins."""
    base = workloads.soup_scene(3000).triangles
    t = np.concatenate([base, base[rng.integers(0, base.size, 3000)], base[:500]])
    t = t[rng.permutation(t.size)]
    nodes = core.bvh_build(t).copy()
    inner = np.flatnonzero(nodes["is_leaf"] == 0)
    pick = inner[rng.random(inner.size) < 0.4]
    for side in ("child_l_aabb", "child_r_aabb"):
        pad = rng.uniform(0.0, 0.05, size=(pick.size, 3)).astype(np.float32) * (rng.random((pick.size, 3)) < 0.5)
        nodes[side]["min"][pick] -= pad
        pad = rng.uniform(0.0, 0.05, size=(pick.size, 3)).astype(np.float32) * (rng.random((pick.size, 3)) < 0.5)
        nodes[side]["max"][pick] += pad
    return abi.SceneArrays(t, nodes, workloads.soup_scene(8).materials, core.sky_d65(1.0))


def test_a_loose_caller_tree_is_answered_from_its_stored_boxes(oracle, monkeypatch):
    """PT_TRAVERSE_EXACT re-derives the boxes of an odd-depth node's children from THEIR children's boxes, which is
    the stored box only in a tree of exact joins.  pt_scene_create checks a caller's tree (k_check_joins); on a padded tree an
    EXACT call is answered by the frame replay, which reads the stored boxes: EXACT, REPLAY and the oracle walking that same
    tree agree on every ray.  With the check switched off (PTC_TRUST_JOINS=1, a test knob) the scoped kernel answers from the
    derived boxes and parts ways with the recursion on some equal-t ties - which is what the check is for."""
    arr = loose_tree_scene(np.random.default_rng(23))
    rays = core.gen_primary_rays(384, 216)
    want = oracle.trace(oracle.scene(arr), rays)
    assert (want["tri"] != abi.PT_NO_TRIANGLE).mean() > 0.5
    ctx = core.Context(0)
    try:
        sc = ctx.scene(arr)
        for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY):
            ctx.set_traversal(mode)
            assert ctx.trace(sc, rays).tobytes() == want.tobytes(), mode
        ctx.set_traversal(abi.PT_TRAVERSE_MIN_T)          # a minimum-t hit on any valid tree
        got = ctx.trace(sc, rays)
        assert np.array_equal(got["t"], want["t"])
        # the same tree rendered: the integrator's queued launches take the fallback too
        ctx.set_traversal(abi.PT_TRAVERSE_EXACT)
        img = ctx.render(sc, 96, 54, 0, 2, 2, seed=3)
        ref = oracle.render(oracle.scene(arr), 96, 54, 0, 2, 2, seed=3)
        assert ((img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))).all()
    finally:
        ctx.close()
    # a tight tree keeps the scoped kernel: nothing to fall back from (checked by the kernel name in the profiles); and the
    # knob that trusts a loose tree shows the difference the check prevents
    monkeypatch.setenv("PTC_TRUST_JOINS", "1")
    ctx = core.Context(0)
    try:
        ctx.set_traversal(abi.PT_TRAVERSE_EXACT)
        got = ctx.trace(ctx.scene(arr), rays)
        differing = int((got["tri"] != want["tri"]).sum())
        assert np.array_equal(got["t"], want["t"]) and differing > 0, differing
    finally:
        ctx.close()


def test_a_box_with_nan_on_one_side_only_rules_out_the_sign_select_step(oracle):
    """a caller box with NaN in just one of min / max on an axis passes a `min > max` test; the sign-select record
    step would load the NaN plane as `near` where minNum / maxNum drop it.  Such a scene must take the general step (and, its
    boxes not being joins, the replay): every mode equals the oracle walking the same nodes."""
    arr = workloads.soup_scene(2000)
    nodes = arr.nodes.copy()
    inner = np.flatnonzero(nodes["is_leaf"] == 0)
    nodes["child_l_aabb"]["min"][inner[5], 0] = np.nan
    nodes["child_r_aabb"]["max"][inner[40], 2] = np.nan
    arr2 = abi.SceneArrays(arr.triangles, nodes, arr.materials, arr.sky)
    rays = core.gen_primary_rays(256, 144)
    want = oracle.trace(oracle.scene(arr2), rays)
    ctx = core.Context(0)
    try:
        sc = ctx.scene(arr2)
        for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY):
            ctx.set_traversal(mode)
            assert ctx.trace(sc, rays).tobytes() == want.tobytes(), mode
    finally:
        ctx.close()
