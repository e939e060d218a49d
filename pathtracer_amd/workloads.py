"""Synthetic workloads; the reference ships no scene.

Everything here is host-side description; triangles come from pt_gen_soup (counter-based RNG), the
node vector from pt_bvh_build (BVH::build mirror).
"""
import numpy as np

from . import abi, core

SOUP_SEED = 0x50554F53  # "SOUP"

# materials 1..8 of the soup: seven flat Lambert colours + one AbsorbedD65 emitter (Material 0 is
# Material::default, scene.rs:204)
_SOUP_COLOURS = [
    (0.75, 0.25, 0.25), (0.25, 0.75, 0.25), (0.25, 0.25, 0.75), (0.75, 0.75, 0.25),
    (0.25, 0.75, 0.75), (0.75, 0.25, 0.75), (0.60, 0.60, 0.60),
]


def soup_materials():
    mats = [abi.default_material()]
    for rgb in _SOUP_COLOURS:
        mats.append(abi.flat_material(rgb))
    mats.append(abi.flat_material((0.8, 0.8, 0.8), emit_rgb=(1.0, 0.9, 0.7)))
    return np.array(mats, dtype=abi.MATERIAL)


def soup_triangles(n, seed=SOUP_SEED):
    """soup: centres in [-1,1]^2 x [-4,-2], edge scale = cube side / n^(1/3) (0.02 at n = 1e6)."""
    s = 2.0 / (float(n) ** (1.0 / 3.0))
    return core.gen_soup(seed, n, (-1.0, -1.0, -4.0), (1.0, 1.0, -2.0), s, 0.0, 8)


def soup20m_triangles(n=20_000_000, seed=SOUP_SEED + 1):
    """config 5: centres in [-4,4]^2 x [-12,-4], edge scale log-uniform in [0.002, 0.2], all diffuse."""
    return core.gen_soup(seed, n, (-4.0, -4.0, -12.0), (4.0, 4.0, -4.0), 0.002, 0.2, 7)


def hw_sky():
    """config 3 sky: turbidity 3, albedo 0.3, zenith +Y, sun normalize(0.3, 0.8, 0.5)."""
    return core.sky_hosek_wilkie(3.0, 0.3, (0.0, 1.0, 0.0), (0.3, 0.8, 0.5))


def soup_scene(n, sky=None, seed=SOUP_SEED, nodes=None):
    tris = soup_triangles(n, seed)
    if nodes is None:
        nodes = core.bvh_build(tris)
    return abi.SceneArrays(tris, nodes, soup_materials(), sky if sky is not None else core.sky_d65(0.0))


def box_triangles():
    """"box-12": a 2x2x2 room centred at z=-3, open towards the camera (5 walls = 10 triangles)
    plus a 2-triangle emitter under the ceiling."""
    x0, x1, y0, y1, z0, z1 = -1.0, 1.0, -1.0, 1.0, -4.0, -2.0
    v = lambda x, y, z: (x, y, z)
    quads = [
        (v(x0, y0, z0), v(x1, y0, z0), v(x1, y1, z0), v(x0, y1, z0), 1),   # back
        (v(x0, y0, z1), v(x0, y0, z0), v(x0, y1, z0), v(x0, y1, z1), 2),   # left
        (v(x1, y0, z0), v(x1, y0, z1), v(x1, y1, z1), v(x1, y1, z0), 3),   # right
        (v(x0, y0, z1), v(x1, y0, z1), v(x1, y0, z0), v(x0, y0, z0), 4),   # floor
        (v(x0, y1, z0), v(x1, y1, z0), v(x1, y1, z1), v(x0, y1, z1), 4),   # ceiling
        (v(-0.3, 0.98, -3.3), v(0.3, 0.98, -3.3), v(0.3, 0.98, -2.7), v(-0.3, 0.98, -2.7), 5),  # emitter
    ]
    tris = np.zeros(2 * len(quads), dtype=abi.TRIANGLE)
    for q, (a, b, c, d, m) in enumerate(quads):
        # quads split (a,b,c), (a,c,d) as scene.rs:258-261
        for k, (p0, p1, p2) in enumerate(((a, b, c), (a, c, d))):
            t = tris[2 * q + k]
            t["a"], t["b"], t["c"] = p0, p1, p2
            t["material_index"] = m
    return tris


def box_materials():
    return np.array([
        abi.default_material(),
        abi.flat_material((0.7, 0.7, 0.7)), abi.flat_material((0.7, 0.2, 0.2)),
        abi.flat_material((0.2, 0.7, 0.2)), abi.flat_material((0.75, 0.75, 0.75)),
        abi.flat_material((0.8, 0.8, 0.8), emit_rgb=(1.0, 1.0, 1.0)),
    ], dtype=abi.MATERIAL)


def box_scene(sky=None):
    tris = box_triangles()
    return abi.SceneArrays(tris, core.bvh_build(tris), box_materials(),
                           sky if sky is not None else core.sky_d65(1.0))


def empty_scene():
    """The reference's default (empty TOML) scene: no meshes, Sky::D65{power: 0}."""
    return abi.SceneArrays(np.zeros(0, dtype=abi.TRIANGLE), None,
                           np.array([abi.default_material()], dtype=abi.MATERIAL), core.sky_d65(0.0))
