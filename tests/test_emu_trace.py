"""The traversal kernels' own SOURCE, run on the CPU: tests/cpp/emu_trace.cpp compiles pathtracer_amd/csrc/trace_kernels.h
(ray tickets and refill, the LDS ring stack over its HBM spill column, EXACT's scope minima and redo list, the frame replay,
MIN_T's global-best pruning) with one fiber per lane and a scheduler that resolves the wave collectives and interleaves a
block's four waves at random, and makes the launcher's sequence of launches.  EXACT and REPLAY must equal the oracle's
traverse_recursive (bvh.rs:250-338) bit for bit on every ray, under every schedule tried; and, built with the host's address
sanitizer, no LDS or global access may leave its array (the buffers have the launcher's exact sizes).  What the GPU adds -
memory ordering inside a wave's stretch between two collectives, timing - is the GPU suite's."""
import ctypes as C
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from pathtracer_amd import abi, core, workloads
from tests import fuzz_scenes

HERE = os.path.dirname(os.path.abspath(__file__))
MIN_T, EXACT, REPLAY = abi.PT_TRAVERSE_MIN_T, abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY


def load(name):
    path = os.path.join(HERE, "cpp", name)
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), name])
    L = C.CDLL(path)
    L.pte_trace.restype = C.c_int
    L.pte_trace.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                            C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32, C.c_uint32]
    return L


def run(L, arr, rays, start=None, mode=EXACT, blocks=8, refill=16, ticket=64, seed=1, queue=None, ring=12):
    rays = np.ascontiguousarray(rays, dtype=abi.RAY)
    nodes = arr.nodes if arr.nodes is not None else np.zeros(0, dtype=abi.BVH_NODE)
    out = np.zeros(rays.size, dtype=abi.HIT)
    stats = (C.c_uint64 * 2)()
    st = None if start is None else np.ascontiguousarray(start, dtype=np.uint32)
    rc = L.pte_trace(mode, abi.ptr(nodes), nodes.size, abi.ptr(arr.triangles), arr.triangles.size, abi.ptr(rays), abi.ptr(st),
                     rays.size, abi.ptr(out), blocks, refill, ticket, seed, stats, abi.ptr(queue), 0 if queue is None else queue.size, ring)
    assert rc == 0, "the emulated launch did not end, or its lanes disagreed on a collective (see stderr)"
    return out, int(stats[1])


def check(L, oracle, arr, rays, start=None, what="", **kw):
    want = oracle.trace(oracle.scene(arr), rays, start)
    got, n_redo = run(L, arr, rays, start, **kw)
    bad = np.flatnonzero((got.view(np.uint8).reshape(rays.size, 16) != want.view(np.uint8).reshape(rays.size, 16)).any(axis=1))
    assert bad.size == 0, f"{what}: {bad.size} rays differ, first {bad[:3]}: got {got[bad[:3]]} want {want[bad[:3]]}"
    return n_redo


@pytest.fixture(scope="module")
def emu():
    return load("libemutrace.so")


@pytest.mark.parametrize("mode", [EXACT, REPLAY])
def test_soup_primaries_and_bounces(emu, oracle, mode):
    arr = workloads.soup_scene(20_000)
    rays = core.gen_primary_rays(160, 90)
    assert check(emu, oracle, arr, rays, what="primaries", mode=mode) == 0
    want = oracle.trace(oracle.scene(arr), rays)
    from tests.golden import fixture_inputs as fi
    brays, start = fi.bounce_rays(rays, want)
    n_redo = check(emu, oracle, arr, brays, start, what="bounces", mode=mode, seed=7)
    assert n_redo < 0.01 * brays.size


@pytest.mark.parametrize("seed", range(4))
def test_schedules_refill_thresholds_and_grids(emu, oracle, seed):
    """The same batch under different wave schedules, refill thresholds (1: a lane takes a new ray the moment it is free; 64: a
    wave refills only when all of it is), ticket sizes and grids (8 blocks: the tickets are used; 64: every ray is in a wave's
    static first block)."""
    arr = workloads.soup_scene(5_000)
    rays = core.gen_primary_rays(128, 72)
    check(emu, oracle, arr, rays, what=f"schedule {seed}", seed=1000 + seed, refill=(1, 16, 32, 64)[seed], ticket=(64, 1, 1024, 64)[seed],
          blocks=(8, 16, 8, 64)[seed])


def stacked_chain(n):
    """n large triangles one behind the other (triangle k at z = -2 - 0.01 (n - k): the LATER ones are nearer), hung on a
    right-leaning chain (node 2k = {left: leaf k, right: the rest}): a ray down -z enters the rest first at every node and
    leaves leaf k pending, so its stack is as deep as the chain."""
    tris = np.zeros(n, dtype=abi.TRIANGLE)
    z = (-2.0 - 0.01 * (n - np.arange(n))).astype(np.float32)
    tris["a"] = np.stack([np.full(n, -5.0), np.full(n, -5.0), z], axis=1)
    tris["b"] = np.stack([np.full(n, 5.0), np.full(n, -5.0), z], axis=1)
    tris["c"] = np.stack([np.zeros(n), np.full(n, 5.0), z], axis=1)
    nodes = np.zeros(2 * n - 1, dtype=abi.BVH_NODE)
    lo = np.minimum(np.minimum(tris["a"], tris["b"]), tris["c"])
    hi = np.maximum(np.maximum(tris["a"], tris["b"]), tris["c"])
    suf_lo = np.minimum.accumulate(lo[::-1], axis=0)[::-1]
    suf_hi = np.maximum.accumulate(hi[::-1], axis=0)[::-1]
    for k in range(n - 1):
        nd = nodes[2 * k]
        nd["child_l_index"], nd["child_r_index"] = 2 * k + 1, 2 * k + 2
        nd["child_l_aabb"]["min"], nd["child_l_aabb"]["max"] = lo[k], hi[k]
        nd["child_r_aabb"]["min"], nd["child_r_aabb"]["max"] = suf_lo[k + 1], suf_hi[k + 1]
        nodes[2 * k + 1]["is_leaf"], nodes[2 * k + 1]["shape_index"] = 1, k
    nodes[2 * n - 2]["is_leaf"], nodes[2 * n - 2]["shape_index"] = 1, n - 1
    return abi.SceneArrays(tris, nodes, workloads.soup_scene(8).materials, core.sky_d65(1.0))


@pytest.mark.parametrize("ring", [8, 10, 16])
def test_the_other_ring_sizes(emu, oracle, ring):
    """PTC_RING picks an instantiation of the scoped kernel with 8, 10 or 16 LDS slots per lane (default 12): same answers, the
    spill column taking over earlier or later."""
    check(emu, oracle, stacked_chain(120), core.gen_primary_rays(64, 16), what=f"ring {ring} chain", ring=ring)
    arr = workloads.soup_scene(5_000)
    check(emu, oracle, arr, core.gen_primary_rays(96, 54), what=f"ring {ring} soup", ring=ring, seed=ring)


@pytest.mark.parametrize("mode", [EXACT, REPLAY, MIN_T])
def test_deep_chain_spills_the_stack(emu, oracle, mode):
    """A chain far deeper than the 12-slot LDS ring with every leaf pending: the oldest entries go to the HBM column and
    come back (the column is exactly as deep as the launcher makes it: one entry more and the sanitizer run below says so)."""
    arr = stacked_chain(200)
    rays = core.gen_primary_rays(64, 32)
    check(emu, oracle, arr, rays, what="chain", mode=mode)


@pytest.mark.parametrize("seed", list(range(12)) + [910, 1912, 8824])
def test_fuzz_scenes(emu, oracle, seed):
    """The sweep's scenes (ties, duplicates, flat boxes, three scales) and rays, 40 % of them axis-parallel (an infinite inverse direction: the general
    form of the record step instead of the sign-select form): 910 / 1912 / 8824 are the seeds round 1's kernels got wrong."""
    rng = np.random.default_rng(seed)
    tris, mats, sky, scale, images = fuzz_scenes.random_scene(rng)
    try:
        nodes = core.bvh_build(tris)
    except core.PtError:
        pytest.skip("degenerate scene: BVH::build fails in the reference too")
    arr = abi.SceneArrays(tris, nodes, mats, sky, images)
    rays = fuzz_scenes.random_rays(rng, 6000, scale)
    want = oracle.trace(oracle.scene(arr), rays)
    check(emu, oracle, arr, rays, what=f"seed {seed}", seed=seed)
    start = np.where(rng.random(rays.size) < 0.3, want["tri"], abi.PT_NO_TRIANGLE).astype(np.uint32)
    check(emu, oracle, arr, rays, start, what=f"seed {seed} with start", seed=seed + 1, mode=(EXACT, REPLAY)[seed % 2])


def check_underflow(L, oracle, n, mode):
    """Tiny triangles, denormal determinants, huge origins, inf / NaN components (tests/test_gpu_trace.py has the story): rays
    whose verdict takes the sign of a NaN are left out (the oracle marks them: that sign is the compiler's choice of operand
    order even between two x86 builds), NaN compares as NaN."""
    arr, tiny, rays = fuzz_scenes.underflow_scene_and_rays(np.random.default_rng(5), n=n)
    want, nan_sign = oracle.trace_nan_flags(oracle.scene(arr), rays)
    got, n_redo = run(L, arr, rays, mode=mode)
    same = (got["tri"] == want["tri"]) & np.all([(np.isnan(got[k]) & np.isnan(want[k])) | (got[k].view(np.uint32) == want[k].view(np.uint32)) for k in "tuv"], axis=0)
    bad = np.flatnonzero(~same & ~nan_sign)
    assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]])
    assert ((got["tri"] == abi.PT_NO_TRIANGLE) | (got["tri"] < arr.triangles.size)).all()
    return n_redo, int((~np.isfinite(want["t"]) & ~nan_sign).sum())


@pytest.mark.parametrize("mode", [EXACT, REPLAY])
def test_irregular_rays_and_non_finite_hits(emu, oracle, mode):
    n_redo, n_non_finite = check_underflow(emu, oracle, 12000, mode)
    if mode == EXACT:
        assert n_redo >= n_non_finite > 20


def test_the_integrators_queued_form(emu, oracle):
    """render.hip's traversal launches: ray POSITIONS come from a queue whose length the kernel reads from a device word, the
    start triangle rides in the ray's pad0, hits are written by position - and the redo pass (irregular rays here: 5 % get an
    infinite origin component) takes its rays from the scoped kernel's list the same way."""
    arr = workloads.soup_scene(20_000)
    prim = core.gen_primary_rays(160, 90)
    first = oracle.trace(oracle.scene(arr), prim)
    from tests.golden import fixture_inputs as fi
    rays, start = fi.bounce_rays(prim, first)
    rng = np.random.default_rng(3)
    rays = rays.copy()
    rays["o"][rng.random(rays.size) < 0.05, 0] = np.inf
    rays["pad0"] = start.view(np.float32)
    queue = np.ascontiguousarray(rng.permutation(rays.size)[: rays.size * 2 // 3].astype(np.uint32))
    want = oracle.trace(oracle.scene(arr), rays, start)
    got, n_redo = run(emu, arr, rays, None, queue=queue, seed=11)
    assert n_redo > 100
    assert got[queue].tobytes() == want[queue].tobytes()
    rest = np.setdiff1d(np.arange(rays.size), queue)
    assert (got["tri"][rest] == 0xDEADBEEF).all()


def test_min_t_returns_a_minimum_t_hit(emu, oracle):
    """MIN_T promises a hit of minimum t, not the recursion's choice among equals: t must be the oracle's bit for bit, the
    triangle the oracle's except where another triangle has that same t."""
    arr = workloads.soup_scene(20_000)
    rays = core.gen_primary_rays(160, 90)
    want = oracle.trace(oracle.scene(arr), rays)
    got, _ = run(emu, arr, rays, mode=MIN_T)
    assert (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    assert (got["tri"] != want["tri"]).mean() < 1e-3


def test_empty_and_tiny_batches(emu, oracle):
    arr = workloads.soup_scene(300)
    for n in (0, 1, 63, 64, 65, 257):
        rays = core.gen_primary_rays(320, 180)[:n]
        for mode in (EXACT, REPLAY):
            check(emu, oracle, arr, rays, what=f"{n} rays", mode=mode)
    one = abi.SceneArrays(arr.triangles[:1], core.bvh_build(arr.triangles[:1]), arr.materials, arr.sky)      # the root is a leaf
    check(emu, oracle, one, core.gen_primary_rays(32, 18), what="one triangle")


def test_under_the_address_sanitizer():
    """The same kind of launches in a child process with the sanitizer's runtime preloaded (it must come first in the process):
    any LDS or global access outside its array ends the child with a report."""
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no address sanitizer runtime for the ROCm clang")
    env = dict(os.environ, LD_PRELOAD=(rt[0] + " " + os.environ.get("LD_PRELOAD", "")).strip(), ASAN_OPTIONS="detect_leaks=0",
               PYTHONPATH=os.path.dirname(HERE))
    p = subprocess.run([sys.executable, "-m", "tests.test_emu_trace"], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "ERROR: AddressSanitizer" not in p.stderr, p.stderr[-3000:] + p.stdout[-500:]
    assert "sanitized launches ok" in p.stdout


def _sanitized_main():
    from tests import oracle as o
    oracle = o.Oracle()
    L = load("libemutrace_asan.so")
    arr = workloads.soup_scene(3_000)
    rays = core.gen_primary_rays(96, 54)
    for seed, mode in enumerate((EXACT, REPLAY, MIN_T)):
        if mode == MIN_T:
            run(L, arr, rays, mode=mode, seed=seed)
        else:
            check(L, oracle, arr, rays, what=f"asan mode {mode}", seed=seed, refill=(1, 16)[seed], mode=mode)
    for mode in (EXACT, REPLAY, MIN_T):
        check(L, oracle, stacked_chain(200), core.gen_primary_rays(32, 16), what="asan chain", mode=mode)
    check_underflow(L, oracle, 4000, EXACT)
    print("sanitized launches ok")


if __name__ == "__main__":
    _sanitized_main()
