"""Checks run against tests/cpp/libptcore_emu.so - the library's OWN sources (core.hip, trace.hip, trace_noalloc.hip, render.hip,
tonemap.hip) compiled for the CPU behind tests/cpp/emu_rt.h - through the ordinary ctypes binding.  Not collected by pytest:
tests/test_emu_lib.py starts this file in a child process with PTCORE_LIB pointing at that library (the binding picks its
library when it is imported) and, for the sanitizer run, the sanitizer's runtime preloaded.  "Device" pointers are host pointers here.
usage: python -m tests.emu_lib_checks <check> [<check> ...]   (prints `ok <check>` per check)"""
import ctypes as C
import os
import sys

import numpy as np

from pathtracer_amd import abi, core, workloads
from tests import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = orc.Oracle()
MODES = (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY, abi.PT_TRAVERSE_MIN_T)


def same_image(a, b):
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def check_trace():
    """pt_trace_closest (host pointers: chunks, the copy streams) in the three modes; bounce rays with start triangles"""
    ctx = core.Context(0)
    arr = workloads.soup_scene(5000)
    sc = ctx.scene(arr)
    osc = O.scene(arr)
    rays = core.gen_primary_rays(96, 54)
    want = O.trace(osc, rays)
    from tests.golden import fixture_inputs as fi
    brays, start = fi.bounce_rays(rays, want)
    bwant = O.trace(osc, brays, start)
    for mode in MODES:
        ctx.set_traversal(mode)
        got = ctx.trace(sc, rays)
        bgot = ctx.trace(sc, brays, start)
        if mode == abi.PT_TRAVERSE_MIN_T:
            assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all() and (bgot["t"].view(np.uint32) == bwant["t"].view(np.uint32)).all()
        else:
            assert got.tobytes() == want.tobytes() and bgot.tobytes() == bwant.tobytes(), mode
    ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
    # irregular rays and an empty scene
    bad = rays[:512].copy()
    bad["d"][::3, 1] = np.inf
    bad["o"][1::3, 0] = np.nan
    w2, nan_sign = O.trace_nan_flags(osc, bad)
    g2 = ctx.trace(sc, bad)
    ok = (g2["tri"] == w2["tri"]) | nan_sign
    assert ok.all()
    e = workloads.empty_scene()
    assert (ctx.trace(ctx.scene(e), rays[:100])["tri"] == abi.PT_NO_TRIANGLE).all()


def check_trace_dev_and_noalloc():
    """the device-pointer entry points (here: host pointers) and the allocation-free one against each other, twice, the second
    call smaller"""
    ctx = core.Context(0)
    arr = workloads.soup_scene(3000)
    sc = ctx.scene(arr)
    rays = np.ascontiguousarray(core.gen_primary_rays(80, 45))
    want = O.trace(O.scene(arr), rays)
    for n in (rays.size, 1000):
        a = np.zeros(n, dtype=abi.HIT)
        b = np.full(n, 0x55, dtype=np.uint8).view(np.uint8)
        b = np.zeros(n, dtype=abi.HIT)
        ctx.trace_dev(sc, rays.ctypes.data, None, n, a.ctypes.data)
        ctx.trace_dev_noalloc(sc, rays.ctypes.data, None, n, b.ctypes.data)
        assert a.tobytes() == want[:n].tobytes() and b.tobytes() == a.tobytes()


def check_trace_submit_wait():
    """pt_trace_closest_dev_submit / _wait: batches queued on several slots (here they run one after the other), different sizes,
    a slot reused, a wait on an idle slot, refusals"""
    ctx = core.Context(0)
    arr = workloads.soup_scene(3000)
    sc = ctx.scene(arr)
    rays = np.ascontiguousarray(core.gen_primary_rays(80, 45))
    want = O.trace(O.scene(arr), rays)
    outs = []
    for i, n in enumerate((rays.size, 1000, 64, 2500, 1)):
        o = np.zeros(n, dtype=abi.HIT)
        ctx.trace_dev_submit(sc, rays.ctypes.data, None, n, o.ctypes.data, slot=i % abi.PT_TRACE_SLOTS)
        outs.append(o)
    for slot in range(abi.PT_TRACE_SLOTS):
        ctx.trace_dev_wait(slot)
    assert ctx.stats().trace_launches == 1 and ctx.stats().trace_rays == 2500       # slot 3 held one batch
    for o in outs:
        assert o.tobytes() == want[:o.size].tobytes()
    ctx.trace_dev_wait(2)                                                          # idle: returns at once
    assert ctx.stats().trace_launches == 0
    lib = core.lib()
    o = np.zeros(8, dtype=abi.HIT)
    rc = lib.pt_trace_closest_dev_submit(ctx.handle, sc.handle, core.C.c_void_p(rays.ctypes.data), None, 8, core.C.c_void_p(o.ctypes.data), abi.PT_TRACE_SLOTS)
    assert rc == abi.PT_ERR_INVALID and b"slot" in lib.pt_last_error(ctx.handle)
    assert lib.pt_trace_closest_dev_wait(ctx.handle, 99) == abi.PT_ERR_INVALID
    assert lib.pt_trace_closest_dev_submit(ctx.handle, sc.handle, None, None, 8, None, 0) == abi.PT_ERR_INVALID
    ctx.close()


def check_loose_tree():
    """a caller's tree whose stored boxes are not the joins: EXACT answers through the frame replay"""
    rng = np.random.default_rng(23)
    base = workloads.soup_scene(600).triangles
    t = np.concatenate([base, base[rng.integers(0, base.size, 600)], base[:100]])
    t = t[rng.permutation(t.size)]
    nodes = core.bvh_build(t).copy()
    inner = np.flatnonzero(nodes["is_leaf"] == 0)
    pick = inner[rng.random(inner.size) < 0.4]
    for side in ("child_l_aabb", "child_r_aabb"):
        nodes[side]["min"][pick] -= rng.uniform(0.0, 0.05, size=(pick.size, 3)).astype(np.float32)
        nodes[side]["max"][pick] += rng.uniform(0.0, 0.05, size=(pick.size, 3)).astype(np.float32)
    arr = abi.SceneArrays(t, nodes, workloads.soup_scene(8).materials, core.sky_d65(1.0))
    ctx = core.Context(0)
    rays = core.gen_primary_rays(64, 36)
    want = O.trace(O.scene(arr), rays)
    sc = ctx.scene(arr)
    for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY):
        ctx.set_traversal(mode)
        assert ctx.trace(sc, rays).tobytes() == want.tobytes(), mode


def check_render_skies():
    """pt_render, one pass: soup under D65 and under the Hosek-Wilkie sky, the box scene (emission), against the oracle's render"""
    ctx = core.Context(0)
    for name, arr in (("d65", workloads.soup_scene(2000)), ("hw", workloads.soup_scene(2000, sky=workloads.hw_sky())), ("box", workloads.box_scene())):
        img = ctx.render(ctx.scene(arr), 40, 24, 0, 6, 6, 99)
        ref = O.render(O.scene(arr), 40, 24, 0, 6, 6, 99)
        assert same_image(img, ref), name
        s = ctx.stats()
        assert s.paths == 40 * 24 * 6 and s.segments >= s.paths


def check_render_passes_and_sums():
    """more samples than the path budget holds (PTC_PASS_PATHS=65536: three passes in two slots, the next one admitted while the
    last one's tail still runs), and PT_RENDER_SUM over two sample ranges: the same image as one oracle render"""
    assert os.environ.get("PTC_PASS_PATHS") == "65536"
    ctx = core.Context(0)
    arr = workloads.soup_scene(1500, sky=workloads.hw_sky())
    sc, osc = ctx.scene(arr), O.scene(arr)
    w, h, spp = 40, 40, 48
    img = ctx.render(sc, w, h, 0, spp, spp, 7)
    assert ctx.stats().trace_launches > 20
    ref = O.render(osc, w, h, 0, spp, spp, 7)
    assert same_image(img, ref)
    a = ctx.render(sc, 24, 16, 0, 5, 8, 3, mode=abi.PT_RENDER_SUM)
    b = ctx.render(sc, 24, 16, 5, 3, 8, 3, mode=abi.PT_RENDER_SUM)
    ra = O.render(osc, 24, 16, 0, 5, 8, 3, mode=abi.PT_RENDER_SUM)
    rb = O.render(osc, 24, 16, 5, 3, 8, 3, mode=abi.PT_RENDER_SUM)
    assert same_image(a, ra) and same_image(b, rb)


def check_scene_file():
    """examples/cornell (TOML + OBJ/MTL + PNG: textures, a dissolve mask, an emissive lamp) through the library's loader and
    renderer"""
    ls = core.load_scene_file(os.path.join(ROOT, "examples", "cornell", "scene.toml"))
    ctx = core.Context(0)
    img = ctx.render(ctx.scene(ls.arrays), 32, 32, 0, 4, 4, 5, sensor_width=ls.camera["sensor_width"], focal_length=ls.camera["focal_length"])
    ref = O.render(O.scene(ls.arrays), 32, 32, 0, 4, 4, 5, sensor_width=ls.camera["sensor_width"], focal_length=ls.camera["focal_length"])
    assert same_image(img, ref)


def check_tonemap():
    """pt_tonemap (two kernels, a block-level tree reduction) against the numpy restatement, to tests/test_tonemap.py's tolerance"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tonemap_oracle
    rng = np.random.default_rng(1)
    img = rng.uniform(0.0, 3.0, size=(37, 53, 4)).astype(np.float32)
    ctx = core.Context(0)
    rgba, bgra8, (avg, mx, mid) = ctx.tonemap(img)
    r2, b2, (avg2, mx2, mid2) = tonemap_oracle.tonemap(img)
    assert abs(avg - avg2) <= 2e-5 * abs(avg2) and abs(mx - mx2) <= 2e-5 * abs(mx2)
    assert (np.abs(rgba - r2) <= 5e-5 * np.maximum(1.0, np.abs(r2))).all()
    assert (np.abs(bgra8.astype(int) - b2.astype(int)) <= 1).all()


def check_lbvh_and_node_paths():
    """lbvh.hip: the device-side BVH build (read back, the oracle walks that tree), and pt_scene_create's device-side node
    flattening and wide collapse held byte for byte against the host-side loops (PTC_FLATTEN=2 PTC_WIDE_COLLAPSE=2 make scene
    creation fail on any difference)"""
    assert os.environ.get("PTC_FLATTEN") == "2" and os.environ.get("PTC_WIDE_COLLAPSE") == "2"
    ctx = core.Context(0)
    rays = core.gen_primary_rays(64, 36)
    for n in (1, 2, 3, 5, 33, 1000, 4000):
        arr = workloads.soup_scene(n)
        sc = ctx.scene(arr)                      # both node paths, compared inside
        want = O.trace(O.scene(arr), rays)
        assert ctx.trace(sc, rays).tobytes() == want.tobytes(), n
        if n >= 2:
            sc2 = ctx.scene(arr, lbvh=True)
            tree = sc2.bvh()
            w2 = O.trace(O.scene(abi.SceneArrays(arr.triangles, tree, arr.materials, arr.sky)), rays)
            for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY):
                ctx.set_traversal(mode)
                assert ctx.trace(sc2, rays).tobytes() == w2.tobytes(), (n, mode)
            ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)


def check_errors():
    """what the library refuses: a child index that points backwards (the device-side validation), a triangle named by two
    leaves, a sky kind that does not exist"""
    ctx = core.Context(0)
    arr = workloads.soup_scene(50)
    inner = np.flatnonzero(arr.nodes["is_leaf"] == 0)
    leaves = np.flatnonzero(arr.nodes["is_leaf"] != 0)
    for what in ("backward", "twice"):
        nodes = arr.nodes.copy()
        if what == "backward":
            nodes["child_l_index"][inner[1]] = 0
        else:
            nodes["shape_index"][leaves[0]] = nodes["shape_index"][leaves[1]]
        try:
            ctx.scene(abi.SceneArrays(arr.triangles, nodes, arr.materials, arr.sky))
            raise AssertionError("accepted: " + what)
        except core.PtError as e:
            assert e.code == abi.PT_ERR_INVALID, what


CHECKS = {k[len("check_"):]: v for k, v in list(globals().items()) if k.startswith("check_")}

if __name__ == "__main__":
    assert "emu" in os.path.basename(os.environ.get("PTCORE_LIB", "")), "PTCORE_LIB must name the CPU build of the library"
    for name in sys.argv[1:]:
        CHECKS[name]()
        print("ok", name, flush=True)
