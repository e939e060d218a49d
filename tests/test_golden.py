"""Committed fixtures (tests/golden/, made by make_fixtures.py with the CPU oracle).

CPU half (not gpu): the oracle still reproduces them — a regression pin of the restatement.
GPU half (gpu): the HIP path, through the C ABI, reproduces them; on the GPU box /root/reference
and a second oracle run are not needed for these.
"""
import json
import os

import numpy as np
import pytest

from pathtracer_amd import abi, core, workloads
from tests.golden import fixture_inputs as fi

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def soup1m():
    return workloads.soup_scene(1_000_000)


@pytest.fixture(scope="module")
def golden_hits():
    z = np.load(os.path.join(G, "soup1m_hits_65536.npz"))
    return z["primary"].view(abi.HIT), z["bounce"].view(abi.HIT), z["bounce_start"]


def same_hits(got, want):
    assert np.array_equal(got["tri"], want["tri"])                       # ids bit-exact (north_star)
    hit = want["tri"] != abi.PT_NO_TRIANGLE
    rel = np.abs(got["t"][hit] - want["t"][hit]) / np.abs(want["t"][hit])
    assert (rel <= 1e-5).all()                                           # t within 1e-5 relative
    assert got.tobytes() == want.tobytes()                               # in fact: every bit


# ---- CPU: oracle vs fixtures -----------------------------------------------------------------------

def test_oracle_reproduces_hit_fixture(oracle, soup1m, golden_hits):
    hp, hb, bstart = golden_hits
    sc = oracle.scene(soup1m)
    prim = fi.primary_subset(1920, 1080, 65536)
    same_hits(oracle.trace(sc, prim), hp)
    brays, start = fi.bounce_rays(prim, hp)
    assert np.array_equal(start, bstart)
    same_hits(oracle.trace(sc, brays, start), hb)


def test_oracle_reproduces_visit_counts(oracle, soup1m):
    g = json.load(open(os.path.join(G, "soup1m_primary_counts.json")))
    rays = core.gen_primary_rays(1920, 1080)
    hits, cnt = oracle.trace(oracle.scene(soup1m), rays, count=True)
    assert cnt[0] / rays.size == g["n_node"] and cnt[1] / rays.size == g["n_tri"]
    assert fi.checksum(hits) == g["hits_checksum"]


def test_oracle_reproduces_images(oracle):
    z = np.load(os.path.join(G, "images.npz"))
    box = oracle.scene(workloads.box_scene())
    assert oracle.render(box, 64, 48, 0, 4, 4, seed=3).tobytes() == z["box_64x48_4spp_seed3"].tobytes()
    soup = oracle.scene(workloads.soup_scene(20_000, sky=workloads.hw_sky()))
    img, cnt = oracle.render(soup, 80, 45, 0, 4, 4, seed=1, count=True)
    assert img.tobytes() == z["soup20k_hw_80x45_4spp_seed1"].tobytes()
    c = json.load(open(os.path.join(G, "images_counts.json")))["soup20k_hw_80x45_4spp_seed1"]
    assert cnt[:2] == [c["paths"], c["segments"]]
    part = oracle.render(soup, 80, 45, 2, 2, 4, seed=1, mode=abi.PT_RENDER_SUM)
    assert part.tobytes() == z["soup20k_hw_80x45_sum_2to4_seed1"].tobytes()


def test_oracle_reproduces_sky_probe(oracle):
    want = np.load(os.path.join(G, "sky_probe.npz"))["radiance"]
    sc = oracle.scene(abi.SceneArrays(np.zeros(0, dtype=abi.TRIANGLE), None,
                                      np.array([abi.default_material()], dtype=abi.MATERIAL), workloads.hw_sky()))
    dirs, wls = fi.sky_probe()
    got = np.stack([oracle.evaluate_sky(sc, d, wls[i % len(wls)]) for i, d in enumerate(dirs)])
    assert got.tobytes() == want.tobytes()
    # sanity of the model itself: below the horizon is black, the disc is far brighter than the sky
    below = dirs[:, 1] <= 0
    assert (got[below] == 0).all()
    assert np.nanmax(got[2]) > 50 * np.nanmax(got[3])


def test_libm_build_of_the_oracle_bounds_the_pt_math_substitution():
    """The reference calls the platform libm; the product and the oracle use include/pt_math.h so
    that CPU and GPU agree bit for bit.  Rendering the same samples with libm instead moves the
    image by <= 1e-5 (mean relative XYZ error) and no pixel by more than 1e-3 relative."""
    from tests.oracle import Oracle
    a, b = Oracle(), Oracle(libm=True)
    assert a.L.pto_uses_libm() == 0 and b.L.pto_uses_libm() == 1
    arr = workloads.soup_scene(20_000, sky=workloads.hw_sky())
    ia = a.render(a.scene(arr), 80, 45, 0, 4, 4, seed=1)
    ib = b.render(b.scene(arr), 80, 45, 0, 4, 4, seed=1)
    assert np.array_equal(np.isnan(ia), np.isnan(ib))
    ia, ib = np.nan_to_num(ia), np.nan_to_num(ib)
    assert np.abs(ia - ib).sum() / np.abs(ib).sum() <= 1e-5
    px = np.abs(ia - ib).max(axis=-1) / np.maximum(np.abs(ib).max(axis=-1), 1e-6)
    assert px.max() <= 1e-3


# ---- GPU: HIP path vs fixtures ----------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("mode", [abi.PT_TRAVERSE_MIN_T, abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY])
def test_gpu_reproduces_hit_fixture(gpu_ctx, soup1m, golden_hits, mode):
    hp, hb, bstart = golden_hits
    sc = gpu_ctx.scene(soup1m)
    gpu_ctx.set_traversal(mode)
    prim = fi.primary_subset(1920, 1080, 65536)
    same_hits(gpu_ctx.trace(sc, prim), hp)
    brays, start = fi.bounce_rays(prim, hp)
    same_hits(gpu_ctx.trace(sc, brays, start), hb)
    gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_MIN_T])
def test_gpu_full_size_batch_checksum(gpu_ctx, soup1m, mode):
    """BASELINE configs[1] at full size: all 2 073 600 primary hits, CRC32 against the oracle's."""
    g = json.load(open(os.path.join(G, "soup1m_primary_counts.json")))
    sc = gpu_ctx.scene(soup1m)
    rays = core.gen_primary_rays(1920, 1080)
    gpu_ctx.set_traversal(mode)
    hits = gpu_ctx.trace(sc, rays)
    assert fi.checksum(hits) == g["hits_checksum"]
    assert abs((hits["tri"] != abi.PT_NO_TRIANGLE).mean() - g["hit_fraction"]) < 1e-12
    # size-independent properties: order independence (a permuted batch gives the permuted hits) ...
    perm = np.random.default_rng(0).permutation(rays.size)
    assert gpu_ctx.trace(sc, rays[perm]).tobytes() == hits[perm].tobytes()
    # ... and the hit point is on the triangle: barycentrics in [0,1], u+v <= 1, t > 0
    h = hits[hits["tri"] != abi.PT_NO_TRIANGLE]
    assert (h["t"] > 0).all() and (h["u"] >= 0).all() and (h["v"] >= 0).all() and (h["u"] + h["v"] <= 1 + 1e-6).all()
    gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)


@pytest.mark.gpu
def test_gpu_reproduces_images(gpu_ctx):
    z = np.load(os.path.join(G, "images.npz"))
    box = gpu_ctx.scene(workloads.box_scene())
    assert gpu_ctx.render(box, 64, 48, 0, 4, 4, seed=3).tobytes() == z["box_64x48_4spp_seed3"].tobytes()
    soup = gpu_ctx.scene(workloads.soup_scene(20_000, sky=workloads.hw_sky()))
    assert gpu_ctx.render(soup, 80, 45, 0, 4, 4, seed=1).tobytes() == z["soup20k_hw_80x45_4spp_seed1"].tobytes()
    c = json.load(open(os.path.join(G, "images_counts.json")))["soup20k_hw_80x45_4spp_seed1"]
    st = gpu_ctx.stats()
    assert (st.paths, st.segments) == (c["paths"], c["segments"])
    part = gpu_ctx.render(soup, 80, 45, 2, 2, 4, seed=1, mode=abi.PT_RENDER_SUM)
    assert part.tobytes() == z["soup20k_hw_80x45_sum_2to4_seed1"].tobytes()
