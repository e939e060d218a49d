#!/usr/bin/env python3
"""Randomised parity sweep, GPU against the CPU oracle (GPU box): random scenes (sizes 1..20000, degenerate and
duplicated triangles, flat and huge/small extents), random rays (axis-parallel directions, origins on box faces,
zero components), all traversal modes + the device-built BVH, and small renders with random materials
(dissolve, emitters, nearest-sampled textures with uv outside [0,1], interpolated vertex normals, random Hosek-Wilkie suns, cameras).  Prints one JSON line; every tree (the reference's, and the device-built one read back with pt_scene_get_bvh) is checked against the CPU restatement walking that same tree: EXACT (the default) and REPLAY must agree on every bit; MIN_T may differ only at equal-t ties and, counted separately, where two hits lie within 4 ulp of each other.
usage: tools/fuzz_parity.py [--seeds N] [--first-seed S] [--seed-list a,b,c] [--rays N]
--emu: the same sweep on the CPU against the library compiled for the host (PTCORE_LIB=tests/cpp/libptcore_emu.so, optionally the
sanitizer build with its runtime preloaded): no torch; keep --sizes and --rays small."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pathtracer_amd import abi, core, workloads
from tests.fuzz_scenes import random_rays, random_scene
from tests.oracle import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=40)
ap.add_argument("--sizes", type=int, nargs="*", default=list(__import__("tests.fuzz_scenes", fromlist=["DEFAULT_SIZES"]).DEFAULT_SIZES), help="triangle counts to draw from")
ap.add_argument("--rays", type=int, default=20000)
ap.add_argument("--first-seed", type=int, default=0, help="continue an earlier sweep: seeds first .. first + seeds - 1")
ap.add_argument("--seed-list", default="", help="comma-separated seeds instead of a range (the driver-run slice of the long sweeps: tests/test_gpu_render.py)")
ap.add_argument("--emu", action="store_true", help="CPU build of the library: see the head of this file")
ap.add_argument("--out", default="", help="also write the JSON line to this file")
a = ap.parse_args()
SEEDS = [int(x) for x in a.seed_list.split(",")] if a.seed_list else list(range(a.first_seed, a.first_seed + a.seeds))
if a.emu:
    assert "emu" in os.path.basename(os.environ.get("PTCORE_LIB", "")), "--emu wants PTCORE_LIB to name the CPU build of the library"
else:
    import torch
    torch.cuda.init()      # (torch first: see tests/conftest.py)
ctx, o = core.Context(0), Oracle()
f32 = np.float32
stats = {"scenes": 0, "rays": 0, "tie_id_mismatches": {}, "t_mismatches": 0, "within_rounding_mismatches": {}, "tie_examples": [], "lbvh_vs_reference_tree": {"rays": 0, "t_differs": 0, "max_ulps": 0, "id_differs_at_equal_t": 0, "hit_in_one_tree_only": 0}, "images": 0, "pixels_differing": {}, "build_errors_agreed": 0, "failures": []}
NAMES = {abi.PT_TRAVERSE_MIN_T: "min_t", abi.PT_TRAVERSE_REPLAY: "replay", abi.PT_TRAVERSE_EXACT: "exact"}
EXACT_MODES = (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY)


for seed in SEEDS:
    rng = np.random.default_rng(seed)
    tris, mats, sky, scale, images = random_scene(rng, a.sizes)
    try:
        nodes = core.bvh_build(tris)
    except core.PtError as e:
        try:
            o.scene(abi.SceneArrays(tris, None, mats, sky, images))
            stats["failures"].append(f"seed {seed}: library build failed ({e}) but the oracle built")
        except Exception:
            stats["build_errors_agreed"] += 1
        continue
    arr = abi.SceneArrays(tris, nodes, mats, sky, images)
    osc = o.scene(arr)
    stats["scenes"] += 1
    print(f"seed {seed}: {tris.size} triangles", file=sys.stderr, flush=True)
    rays = random_rays(rng, a.rays, scale)
    want = o.trace(osc, rays)
    start = np.where(rng.random(rays.size) < 0.3, want["tri"], abi.PT_NO_TRIANGLE).astype(np.uint32)
    want_s = o.trace(osc, rays, start)
    for lbvh in (False, True):
        if lbvh and tris.size < 2:
            continue
        sc = ctx.scene(arr, lbvh=lbvh)
        if lbvh:
            # the device-built tree is checked like the reference's: read back as `BVH.nodes` (pt_scene_get_bvh) and
            # walked by the CPU restatement of traverse_recursive.  Against the reference-built tree only the topology
            # differs, so hits may differ where two candidates lie within rounding of each other: equal t, or a box
            # entry distance ((min - o) * inv) an ulp or two above the t of the triangle inside it, which
            # `first.distance > far_min` (bvh.rs:285) then prunes in one tree and not in the other.
            tree_osc = o.scene(abi.SceneArrays(tris, sc.bvh(), mats, sky, images))
            tree_want, tree_want_s = o.trace(tree_osc, rays), o.trace(tree_osc, rays, start)
            for x, y in ((tree_want, want), (tree_want_s, want_s)):
                dt = x["t"].view(np.uint32) != y["t"].view(np.uint32)
                stats["lbvh_vs_reference_tree"]["rays"] += rays.size
                stats["lbvh_vs_reference_tree"]["t_differs"] += int(dt.sum())
                stats["lbvh_vs_reference_tree"]["id_differs_at_equal_t"] += int(((x["tri"] != y["tri"]) & ~dt).sum())
                both = dt & (x["tri"] != abi.PT_NO_TRIANGLE) & (y["tri"] != abi.PT_NO_TRIANGLE)
                ulps = np.abs(x["t"].view(np.int32).astype(np.int64) - y["t"].view(np.int32).astype(np.int64))
                stats["lbvh_vs_reference_tree"]["hit_in_one_tree_only"] += int((dt & ~both).sum())
                if both.any():
                    stats["lbvh_vs_reference_tree"]["max_ulps"] = max(stats["lbvh_vs_reference_tree"]["max_ulps"], int(ulps[both].max()))
        else:
            tree_osc, tree_want, tree_want_s = osc, want, want_s
        for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_MIN_T, abi.PT_TRAVERSE_REPLAY):
            ctx.set_traversal(mode)
            if tris.size >= 100000:
                print(f"  lbvh {lbvh} mode {NAMES[mode]}", file=sys.stderr, flush=True)      # big scenes: keep the log moving
            for got, ref in ((ctx.trace(sc, rays), tree_want), (ctx.trace(sc, rays, start), tree_want_s)):
                stats["rays"] += rays.size
                key0 = "lbvh/" if lbvh else "reference-tree/"
                tb = got["t"].view(np.uint32) != ref["t"].view(np.uint32)
                idm = (got["tri"] != ref["tri"]) & ~tb
                if mode not in EXACT_MODES and tb.any():
                    # MIN_T prunes a subtree against the best hit of the WHOLE ray (`best_t > entry`), the recursion
                    # only against the near sibling's result (bvh.rs:285).  In exact arithmetic both keep the minimum; in
                    # binary32 a box's entry distance ((min - o) * inv) can lie an ulp or two ABOVE the distance of the
                    # triangle inside it (Havel-Herout), so a hit from another subtree that close can cut it.  Counted,
                    # bounded to 4 ulp, not a failure; EXACT and REPLAY have no such case.
                    near = tb & (got["tri"] != abi.PT_NO_TRIANGLE) & (ref["tri"] != abi.PT_NO_TRIANGLE) & \
                        (np.abs(got["t"].view(np.int32).astype(np.int64) - ref["t"].view(np.int32).astype(np.int64)) <= 4)
                    stats["within_rounding_mismatches"][key0 + NAMES[mode]] = stats["within_rounding_mismatches"].get(key0 + NAMES[mode], 0) + int(near.sum())
                    tb &= ~near
                stats["t_mismatches"] += int(tb.sum())
                key = ("lbvh/" if lbvh else "reference-tree/") + NAMES[mode]
                stats["tie_id_mismatches"][key] = stats["tie_id_mismatches"].get(key, 0) + int(idm.sum())
                if idm.any() and len(stats["tie_examples"]) < 8:
                    i = int(np.flatnonzero(idm)[0])
                    stats["tie_examples"].append(f"seed {seed} {key}: ray {i} got {got[i]} want {ref[i]}")
                if idm.any() and mode in EXACT_MODES:
                    i = int(np.flatnonzero(idm)[0])
                    stats["failures"].append(f"seed {seed} lbvh {lbvh}: {NAMES[mode]} must reproduce ties too: ray {i} got {got[i]} want {ref[i]}")
                if tb.any():
                    i = int(np.flatnonzero(tb)[0])
                    stats["failures"].append(f"seed {seed} lbvh {lbvh} mode {mode}: ray {i} got {got[i]} want {ref[i]}")
        ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
        if seed % 2 == 0:
            fl, sw = float(rng.uniform(20, 120)), float(rng.uniform(10, 50))
            rmode = abi.PT_RENDER_FINAL | (abi.PT_RENDER_NORMALIZE_SKY_DIRECTION if seed % 4 == 2 else 0)
            ref = o.render(tree_osc, 48, 32, 0, 4, 4, seed=seed, sensor_width=sw, focal_length=fl, mode=rmode)
            for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_MIN_T):
                ctx.set_traversal(mode)
                img = ctx.render(sc, 48, 32, 0, 4, 4, seed=seed, sensor_width=sw, focal_length=fl, mode=rmode)
                diff = ~((img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))).all(axis=-1)
                key = ("lbvh/" if lbvh else "reference-tree/") + NAMES[mode]
                stats["pixels_differing"][key] = stats["pixels_differing"].get(key, 0) + int(diff.sum())
                if diff.any() and mode == abi.PT_TRAVERSE_EXACT:
                    stats["failures"].append(f"seed {seed} lbvh {lbvh}: EXACT render differs in {int(diff.sum())} pixels")
            ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
            stats["images"] += 1
stats["ok"] = not stats["failures"] and stats["t_mismatches"] == 0
stats["failures"] = stats["failures"][:10]
if a.emu:
    stats["library"] = os.path.basename(os.environ["PTCORE_LIB"])
print(json.dumps(stats))
if a.out:
    json.dump(stats, open(a.out, "w"), indent=1)
sys.exit(0 if stats["ok"] else 1)
