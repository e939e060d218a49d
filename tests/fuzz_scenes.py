"""Random scenes and rays of the parity sweep (tools/fuzz_parity.py) - also used by the tests that keep the sweep's
two instructive rays (tests/test_gpu_trace.py).  Scene `seed` = random_scene(default_rng(seed)) followed by
random_rays on the same generator: do not reorder the draws, the seeds quoted in DESIGN.md depend on them."""
import numpy as np

from pathtracer_amd import abi, core, workloads

f32 = np.float32
DEFAULT_SIZES = (1, 2, 3, 7, 64, 500, 5000, 20000)


def random_scene(rng, sizes=DEFAULT_SIZES):
    n = int(rng.choice(sizes))
    scale = f32(rng.choice([1e-3, 1.0, 1e3]))
    c = rng.uniform(-1, 1, size=(n, 3)).astype(f32)
    c[:, 2] -= 3
    e1, e2 = (rng.uniform(-0.3, 0.3, size=(n, 3)).astype(f32) for _ in range(2))
    kind = rng.integers(0, 6, size=n)
    e2[kind == 0] = e1[kind == 0] * f32(2)                  # zero-area (collinear)
    e1[kind == 1, 2] = 0; e2[kind == 1, 2] = 0              # axis-aligned flat triangles
    t = np.zeros(n, dtype=abi.TRIANGLE)
    t["a"], t["b"], t["c"] = c * scale, (c + e1) * scale, (c + e2) * scale
    dup = kind == 2
    if dup.sum() > 3000:                                     # a cluster of k copies costs every ray through it k triangle tests
        dup[np.flatnonzero(dup)[3000:]] = False
    if dup.sum() > 1:                                        # exact duplicates of one triangle
        src = np.flatnonzero(dup)[0]
        for k in "abc":
            t[k][dup] = t[k][src]
    t["material_index"] = rng.integers(0, 7, size=n)
    # uv (also outside [0,1]: rem_euclid) and un-normalised vertex normals on random subsets
    has_uv, has_n = rng.random(n) < 0.5, rng.random(n) < 0.3
    t["uv"] = rng.uniform(-2, 3, size=(n, 3, 2)).astype(f32)
    t["vertex_normals"] = rng.normal(size=(n, 3, 3)).astype(f32)
    t["flags"] = has_uv * abi.PT_TRI_HAS_UV + has_n * abi.PT_TRI_HAS_NORMALS
    images = [rng.uniform(0, 1, size=(5, 3, 3)).astype(f32), rng.uniform(0, 1, size=(4, 7)).astype(f32),
              rng.uniform(0, 2, size=(1, 1, 3)).astype(f32)]
    tex_diffuse, tex_dissolve, tex_emit = abi.default_material(), abi.default_material(), abi.default_material()
    tex_diffuse["diffuse"]["kind"], tex_diffuse["diffuse"]["image"] = abi.PT_TEX_NEAREST, 0
    tex_dissolve["dissolve"]["kind"], tex_dissolve["dissolve"]["image"] = abi.PT_TEX_NEAREST, 1
    tex_dissolve["base_dissolve"] = 0.9
    tex_emit["emit_kind"] = abi.PT_EMIT_ABSORBED_D65
    tex_emit["absorption"]["kind"], tex_emit["absorption"]["image"] = abi.PT_TEX_NEAREST, 2
    mats = np.array([abi.default_material(),
                     abi.flat_material(rng.uniform(0, 1, 3)),
                     abi.flat_material(rng.uniform(0, 1, 3), emit_rgb=rng.uniform(0, 2, 3)),
                     abi.flat_material(rng.uniform(0, 1, 3), base_dissolve=float(rng.uniform(0, 1))),
                     tex_diffuse, tex_dissolve, tex_emit], dtype=abi.MATERIAL)
    if rng.random() < 0.5:
        sun = rng.normal(size=3)
        sun[1] = abs(sun[1]) + 0.05                             # above the horizon (scene.rs:121)
        sky = core.sky_hosek_wilkie(float(rng.uniform(1, 10)), float(rng.uniform(0, 1)), (0, 1, 0), sun / np.linalg.norm(sun))
    else:
        sky = core.sky_d65(float(rng.uniform(0, 2)))
    return t, mats, sky, scale, images


def random_rays(rng, n, scale):
    r = np.zeros(n, dtype=abi.RAY)
    r["o"] = (rng.uniform(-2, 2, size=(n, 3)) * [1, 1, 1] + [0, 0, 0]).astype(f32) * scale
    d = rng.normal(size=(n, 3)).astype(f32)
    kind = rng.integers(0, 5, size=n)
    d[kind == 0, 0] = 0                                      # zero components: +-inf inverse, NaN slabs
    d[kind == 1, :2] = 0
    d[kind == 2] *= f32(1e-6)                                # tiny, un-normalised
    d[kind == 3] = (np.array([0, 0, -3], dtype=f32) * scale - r["o"][kind == 3])      # aimed at the cloud
    r["d"] = d
    return r


def underflow_scene_and_rays(rng, n=60000):
    """The NaN-sign and overflow classes (tests/test_gpu_trace.py): a 5000-triangle soup in which 400 triangles are shrunk to
    edges of 1e-12 .. 1e-20 around the world origin (|n|^2 underflows: n1, n2 huge, inf or NaN), and six kinds of ray:
    0 ordinary, 1 directions of 1e-37 .. 1e-39 (denormal determinant, 1 / det = inf: hits at distance +inf), 2 origins of
    1e30, 3 origins ON the tiny triangles, 4 an infinite or NaN component or a denormal direction, 5 ordinary."""
    arr = workloads.soup_scene(5000)
    t = arr.triangles.copy()
    tiny = rng.choice(t.size, 400, replace=False)
    # tiny triangles live around the world origin, where binary32 can hold edges of 1e-12 .. 1e-20
    shrink = (10.0 ** rng.uniform(-19, -11, size=(tiny.size, 1))).astype(np.float32)
    e1, e2 = t["b"][tiny] - t["a"][tiny], t["c"][tiny] - t["a"][tiny]
    t["a"][tiny] = rng.uniform(-1, 1, size=(tiny.size, 3)).astype(np.float32) * shrink
    t["b"][tiny] = t["a"][tiny] + e1 * shrink * np.float32(50)
    t["c"][tiny] = t["a"][tiny] + e2 * shrink * np.float32(50)
    arr2 = abi.SceneArrays(t, core.bvh_build(t), arr.materials, arr.sky)
    rays = np.zeros(n, dtype=abi.RAY)
    rays["o"] = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32) + np.float32([0, 0, 1])
    target = t["a"][rng.integers(0, t.size, n)]
    d = (target - rays["o"]) * rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    kind = np.arange(n) % 6
    d[kind == 1] *= (10.0 ** rng.uniform(-39.5, -37.3, size=(int((kind == 1).sum()), 1))).astype(np.float32)   # tiny directions: denormal det, 1/det = inf
    rays["d"] = d
    rays["o"][kind == 2] *= np.float32(1e30)                  # huge coordinates: t overflows
    rays["d"][kind == 2] = rng.normal(size=(int((kind == 2).sum()), 3)).astype(np.float32)
    on = rng.choice(tiny, int((kind == 3).sum()))             # origins ON the tiny triangles (a vertex, or inside by a few parts in 1e7)
    rays["o"][kind == 3] = t["a"][on] + (t["b"][on] - t["a"][on]) * rng.uniform(0, 0.5, size=(on.size, 1)).astype(np.float32) + (t["c"][on] - t["a"][on]) * rng.uniform(0, 0.5, size=(on.size, 1)).astype(np.float32)
    rays["d"][kind == 3] = rng.normal(size=(on.size, 3)).astype(np.float32)
    odd = np.flatnonzero(kind == 4)
    rays["o"][odd[0::6], 0] = np.inf
    rays["o"][odd[1::6], 1] = np.nan
    rays["d"][odd[2::6], 2] = np.inf
    rays["d"][odd[3::6], 0] = -np.inf
    rays["d"][odd[4::6], 1] = np.nan
    rays["d"][odd[5::6]] *= np.float32(1e-41) / np.abs(rays["d"][odd[5::6]]).max(axis=1, keepdims=True)   # denormal directions
    return arr2, tiny, rays
