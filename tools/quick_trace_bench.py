#!/usr/bin/env python3
"""Development aid: time pt_trace_closest_dev on soup-N primary rays in every traversal mode."""
import argparse
import ctypes as C
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pathtracer_amd import abi, core, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--fast-only", action="store_true", help="time the default mode only")
ap.add_argument("--bounce", action="store_true", help="time incoherent secondary rays (tests/golden/fixture_inputs.bounce_rays) instead of primaries")
ap.add_argument("--sort", choices=["none", "origin", "origin+octant", "octant"], default="none",
                help="with --bounce: reorder the secondary rays by a 30-bit Morton key of the origin and/or the direction octant")
ap.add_argument("--device-only", action="store_true", help="skip the host-pointer entry point")
ap.add_argument("--tile", type=int, default=0, help="reorder rays into TxT pixel tiles")
ap.add_argument("--leaf-order", action="store_true", help="experiment: hand the triangles over in the order of the BVH's leaves (rebuilt on the permuted array)")
a = ap.parse_args()

t = time.time()
arr = workloads.soup_scene(a.tris)
if a.leaf_order:
    order = arr.nodes["shape_index"][arr.nodes["is_leaf"] != 0]
    t2 = np.ascontiguousarray(arr.triangles[order])
    arr = abi.SceneArrays(t2, core.bvh_build(t2), arr.materials, arr.sky)
print(f"scene built in {time.time()-t:.2f}s", flush=True)
ctx = core.Context(0)
sc = ctx.scene(arr)
i = sc.info()
print("info", i.n_triangles, i.n_internal_nodes, i.max_depth, i.device_bytes, flush=True)
rays = core.gen_primary_rays(a.width, a.height)
if a.tile:
    T = a.tile
    ys, xs = np.mgrid[0:a.height, 0:a.width]
    key = ((ys // T) * ((a.width + T - 1) // T) + (xs // T)) * (T * T) + (ys % T) * T + (xs % T)
    rays = rays[np.argsort(key.ravel(), kind="stable")]
d_start_ptr = 0
if a.bounce:
    from tests.golden import fixture_inputs as fi
    first = ctx.trace(sc, rays)
    rays, start = fi.bounce_rays(rays, first)
    if a.sort != "none":
        o = rays["o"]
        q = np.clip(((o - o.min(axis=0)) / (np.ptp(o, axis=0) + 1e-20) * 1023).astype(np.uint64), 0, 1023)

        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        morton = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
        octant = ((rays["d"][:, 0] < 0).astype(np.uint64) << 2) | ((rays["d"][:, 1] < 0).astype(np.uint64) << 1) | (rays["d"][:, 2] < 0).astype(np.uint64)
        key = {"origin": morton, "origin+octant": (octant << 30) | morton, "octant": octant}[a.sort]
        order = np.argsort(key, kind="stable")
        rays, start = np.ascontiguousarray(rays[order]), np.ascontiguousarray(start[order])
    d_start = torch.from_numpy(start.view(np.int32)).cuda()
    d_start_ptr = d_start.data_ptr()
n = rays.size
d_rays = torch.from_numpy(rays.view(np.float32).reshape(n, 8)).cuda()
d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
res = {}
for mode, name in ((abi.PT_TRAVERSE_MIN_T, "min_t"), (abi.PT_TRAVERSE_EXACT, "exact"), (abi.PT_TRAVERSE_REPLAY, "replay")):
    if a.fast_only and name != "exact":
        continue
    ctx.set_traversal(mode)
    ms = []
    for r in range(a.reps + 2):
        ctx.trace_dev(sc, d_rays.data_ptr(), d_start_ptr, n, d_out.data_ptr())
        ms.append(ctx.stats().trace_ms)
    ms = ms[2:]
    res[name] = d_out.cpu().numpy().copy()
    print(f"{name}: {np.mean(ms):.3f} ms (min {np.min(ms):.3f}) -> {n/np.mean(ms)/1e3:.1f} Mrays/s", flush=True)
if not a.bounce and not a.device_only:
    ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
    out = np.zeros(n, dtype=abi.HIT)
    ctx.trace(sc, rays, out=out)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.trace(sc, rays, out=out)
    dt = (time.perf_counter() - t0) / 5
    print(f"host-pointer pt_trace_closest (upload + kernel + download, chunked), pageable numpy buffers: {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} Mrays/s", flush=True)
    p_rays, p_out = core.PinnedArray(n, abi.RAY), core.PinnedArray(n, abi.HIT)
    p_rays.array[:] = rays
    for _ in range(6):          # the first calls on fresh pinned pages are slower
        ctx.trace(sc, p_rays.array, out=p_out.array)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.trace(sc, p_rays.array, out=p_out.array)
    dt = (time.perf_counter() - t0) / 5
    print(f"host-pointer pt_trace_closest, pinned buffers (pt_host_alloc): {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} Mrays/s; "
          f"device side {ctx.stats().total_ms:.2f} ms, kernels {ctx.stats().trace_ms:.2f} ms in {ctx.stats().trace_launches} launches; "
          f"same hits: {p_out.array.tobytes() == out.tobytes()} crc {__import__('zlib').crc32(out.tobytes())}", flush=True)
import zlib
print({k: zlib.crc32(v.tobytes()) for k, v in res.items()}, "all equal:", all(np.array_equal(v, res["exact"]) for v in res.values()))
