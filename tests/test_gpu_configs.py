"""BASELINE configs[2], [3], [4] through the C ABI on the GPU, against fixtures the CPU oracle
made in the build container (tests/golden/make_fixtures_big.py; provenance and "parity unpinned" caveat there):

  config 3  soup-1M + Hosek-Wilkie sky at 1920x1080: sample 0 of every pixel of the full frame (checksum + NaN mask size) and
            all 64 spp bit for bit on a 240x135 frame of the same camera - in one pass, and again cut into 32 passes
            (PTC_PASS_PATHS forced small: the multi-pass accumulation order);
  config 4  the same scene at 3840x2160: one sample range of the full frame by checksum, the ranges [0,2) + [2,4) against [0,4)
            (alpha = samples, NaN mask stable across the split, sums equal to f32 rounding) at full size and bit for bit per
            range at 160x90;
  config 5  soup-20M (built on the box: pt_gen_soup + pt_bvh_build, node vector checked by checksum) at 3840x2160: all
            8 294 400 primary hits by checksum, 2 x 32 768 primary / bounce hits and a small render bit for bit.

Images are compared with NaN == NaN (the reference's un-normalised sky direction poisons most pixels of these scenes,
DESIGN.md 2)."""
import json
import os

import numpy as np
import pytest

from pathtracer_amd import abi, core, workloads
from tests.golden import fixture_inputs as fi

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(G, "configs345.json"))), np.load(os.path.join(G, "configs345.npz"))


@pytest.fixture(scope="module")
def soup1m_hw(gpu_ctx):
    arr = workloads.soup_scene(1_000_000, sky=workloads.hw_sky())
    sc = gpu_ctx.scene(arr)
    yield arr, sc
    sc.close()


def same_image(got, want):
    return np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(want).view(np.uint32))


def test_config3_full_frame_sample_and_64spp_image(gpu_ctx, soup1m_hw, golden):
    meta, npz = golden
    arr, sc = soup1m_hw
    m = meta["config3_1080p_sum_sample0"]
    img = gpu_ctx.render(sc, 1920, 1080, m["spp_begin"], m["spp_count"], m["spp_total"], seed=m["seed"], mode=abi.PT_RENDER_SUM)
    assert int(np.isnan(img).any(axis=-1).sum()) == m["nan_pixels"]
    assert fi.image_checksum(img) == m["crc"]
    assert (img[..., 3] == 1.0).all()
    want = npz["config3_240x135_64spp"]
    got = gpu_ctx.render(sc, 240, 135, 0, 64, 64, seed=1)
    assert same_image(got, want)
    st = gpu_ctx.stats()
    assert st.paths == 240 * 135 * 64 and st.segments > 3 * st.paths


@pytest.mark.parametrize("group", [1, 2, 12, 64])
def test_config3_image_does_not_depend_on_how_samples_are_grouped(monkeypatch, golden, group):
    """PTC_SAMPLE_GROUP: how many samples of a pixel sit side by side in the path ids (which rays share a traversal wave, and how
    k_accumulate cuts its blocks).  12 is not a power of two and becomes 8."""
    meta, npz = golden
    monkeypatch.setenv("PTC_SAMPLE_GROUP", str(group))
    ctx = core.Context(0)
    try:
        sc = ctx.scene(workloads.soup_scene(1_000_000, sky=workloads.hw_sky()))
        assert same_image(ctx.render(sc, 240, 135, 0, 64, 64, seed=1), npz["config3_240x135_64spp"])
        a = ctx.render(sc, 61, 37, 3, 12, 64, seed=5, mode=abi.PT_RENDER_SUM)        # 12 samples from the 4th on, odd image size
        monkeypatch.setenv("PTC_SAMPLE_GROUP", "1")
        ref = core.Context(0)
        try:
            b = ref.render(ref.scene(workloads.soup_scene(1_000_000, sky=workloads.hw_sky())), 61, 37, 3, 12, 64, seed=5, mode=abi.PT_RENDER_SUM)
        finally:
            ref.close()
        assert same_image(a, b)
        sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("slots,admit", [(1, 25), (2, 25), (3, 25), (4, 100), (3, 0)])
def test_config3_image_is_the_same_when_cut_into_many_passes(monkeypatch, golden, slots, admit):
    """PTC_PASS_PATHS = 129 600 paths in flight: the 64 spp of the 240x135 frame become passes of 4 / slots samples, up to
    `slots` of them in flight together (a pass joins the queue once <= admit % of a pass's paths are live; it is folded into
    the image when its last path has ended, oldest pass first).  Each pixel's samples are still added in sample order, so the
    image is the one-pass image bit for bit, whatever the schedule."""
    meta, npz = golden
    monkeypatch.setenv("PTC_PASS_PATHS", str(4 * 240 * 135))
    monkeypatch.setenv("PTC_PASS_SLOTS", str(slots))
    monkeypatch.setenv("PTC_ADMIT_PERCENT", str(admit))
    ctx = core.Context(0)
    try:
        arr = workloads.soup_scene(1_000_000, sky=workloads.hw_sky())
        sc = ctx.scene(arr)
        got = ctx.render(sc, 240, 135, 0, 64, 64, seed=1)
        assert same_image(got, npz["config3_240x135_64spp"])
        st = ctx.stats()
        assert st.paths == 240 * 135 * 64
        per_pass = st.trace_launches / (64 // max(1, 4 // slots))
        print(f"slots {slots} admit {admit}: {st.trace_launches} launches, {per_pass:.1f} per pass")
        if slots == 1 or admit == 0:
            assert per_pass > 12            # one pass at a time: every pass runs its own tail of thin iterations
        else:                               # passes overlap: the tail of one rides in the launches of the next
            assert per_pass < 24 / slots + 1.5
        sc.close()
    finally:
        ctx.close()


def test_short_of_memory_the_pass_is_halved_not_the_image(monkeypatch, golden):
    """When the workspace for the path budget does not fit, pt_render halves the pass until it does (one sample of every
    pixel is the floor, then fewer slots), and fails with PT_ERR_NOMEM only when not even that fits.  PTC_WORK_LIMIT_MB
    stands in for an HBM that is nearly full: 64 spp of 240x135 want 2.3 GB in one pass."""
    meta, npz = golden
    arr = workloads.soup_scene(1_000_000, sky=workloads.hw_sky())
    for limit_mb, ok in ((256, True), (80, True), (8, False)):
        monkeypatch.setenv("PTC_WORK_LIMIT_MB", str(limit_mb))
        ctx = core.Context(0)
        try:
            sc = ctx.scene(arr)
            if ok:
                got = ctx.render(sc, 240, 135, 0, 64, 64, seed=1)
                assert same_image(got, npz["config3_240x135_64spp"]), limit_mb
                assert ctx.stats().paths == 240 * 135 * 64 and ctx.stats().trace_launches > 2 * 18        # several passes
            else:
                with pytest.raises(core.PtError) as e:
                    ctx.render(sc, 240, 135, 0, 64, 64, seed=1)
                assert e.value.code == abi.PT_ERR_NOMEM
                assert ctx.trace(sc, core.gen_primary_rays(16, 16)).size == 256                         # the context is still usable
            sc.close()
        finally:
            ctx.close()


def test_config4_sample_ranges_at_3840x2160(gpu_ctx, soup1m_hw, golden):
    meta, npz = golden
    arr, sc = soup1m_hw
    m = meta["config4_2160p_sum_sample3of4"]
    one = gpu_ctx.render(sc, 3840, 2160, m["spp_begin"], m["spp_count"], m["spp_total"], seed=m["seed"], mode=abi.PT_RENDER_SUM)
    assert int(np.isnan(one).any(axis=-1).sum()) == m["nan_pixels"] and fi.image_checksum(one) == m["crc"]
    # the split of config 4: ranges [0,2) and [2,4) of a 4-spp render, as two ranks would make them
    a = gpu_ctx.render(sc, 3840, 2160, 0, 2, 4, seed=1, mode=abi.PT_RENDER_SUM)
    b = gpu_ctx.render(sc, 3840, 2160, 2, 2, 4, seed=1, mode=abi.PT_RENDER_SUM)
    full = gpu_ctx.render(sc, 3840, 2160, 0, 4, 4, seed=1, mode=abi.PT_RENDER_SUM)
    assert (a[..., 3] == 2.0).all() and (b[..., 3] == 2.0).all() and (full[..., 3] == 4.0).all()
    s = a + b
    assert np.array_equal(np.isnan(s), np.isnan(full))                  # a NaN sample poisons its pixel either way
    ok = ~np.isnan(full)
    # (s0 + s1) + (s2 + s3) against ((s0 + s1) + s2) + s3: 32 non-negative terms per channel (8 wavelengths x 4 samples) in two association orders: equal to a few f32 roundings
    assert np.allclose(s[ok], full[ok], rtol=4e-6, atol=0.0)
    # sample 3 alone (the checksummed buffer above) and sample 2 alone add up to the range [2,4) the same way
    three = gpu_ctx.render(sc, 3840, 2160, 2, 1, 4, seed=1, mode=abi.PT_RENDER_SUM)
    s23 = three + one
    okb = ~np.isnan(b)
    assert np.array_equal(np.isnan(s23), np.isnan(b)) and np.allclose(s23[okb], b[okb], rtol=4e-6, atol=0.0)
    # per range, bit for bit against the oracle at 160x90
    for bgn, cnt in ((0, 2), (2, 2), (0, 4)):
        got = gpu_ctx.render(sc, 160, 90, bgn, cnt, 4, seed=1, mode=abi.PT_RENDER_SUM)
        assert same_image(got, npz[f"config4_160x90_sum_{bgn}to{bgn + cnt}_of4"]), (bgn, cnt)


def test_config5_soup20m_at_3840x2160(gpu_ctx, golden):
    import torch
    meta, npz = golden
    m = meta["config5_soup20m_2160p_primaries"]
    free, total = torch.cuda.mem_get_info()
    if free < 16 << 30:
        pytest.skip("less than 16 GB of HBM free")
    tris = workloads.soup20m_triangles()
    nodes = core.bvh_build(tris)
    assert nodes.size == m["n_nodes"] and fi.checksum(nodes) == m["nodes_crc"]          # the box builds the container's tree
    arr = abi.SceneArrays(tris, nodes, workloads.soup_materials(), workloads.hw_sky())
    sc = gpu_ctx.scene(arr)
    try:
        assert sc.info().n_triangles == 20_000_000
        rays = core.gen_primary_rays(3840, 2160)
        n = m["subset"]
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(rays.size) // np.uint64(n)).astype(np.int64)
        sub = rays[idx].copy()
        hp = npz["config5_primary"]
        brays, bstart = fi.bounce_rays(sub, hp)
        assert np.array_equal(bstart, npz["config5_bounce_start"])
        for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_MIN_T, abi.PT_TRAVERSE_REPLAY):
            gpu_ctx.set_traversal(mode)
            assert gpu_ctx.trace(sc, sub).tobytes() == hp.tobytes(), mode
            assert gpu_ctx.trace(sc, brays, bstart).tobytes() == npz["config5_bounce"].tobytes(), mode
        gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
        hits = gpu_ctx.trace(sc, rays)                                   # the whole 4K batch, default traversal
        assert fi.checksum(hits) == m["hits_checksum"]
        assert abs(float((hits["tri"] != abi.PT_NO_TRIANGLE).mean()) - m["hit_fraction"]) < 1e-12
        assert same_image(gpu_ctx.render(sc, 96, 54, 0, 2, 2, seed=1), npz["config5_96x54_2spp"])
        # a few samples of the 4K frame itself: alpha = samples, finite where the low-res image is mostly finite or not - only
        # the plumbing at this size (8.3 M paths per sample) is checked here, the values above
        img = gpu_ctx.render(sc, 3840, 2160, 0, 2, 16, seed=1, mode=abi.PT_RENDER_SUM)
        assert (img[..., 3] == 2.0).all() and gpu_ctx.stats().paths == 3840 * 2160 * 2
    finally:
        gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
        sc.close()
