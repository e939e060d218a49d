#!/usr/bin/env python3
"""Development aid: time pt_render_dev on soup-N + Hosek-Wilkie sky (BASELINE configs[2] shape)."""
import argparse
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pathtracer_amd import abi, core, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--leaf-order", action="store_true", help="experiment: hand the triangles over in the order of the BVH's leaves (rebuilt on the permuted array)")
a = ap.parse_args()

arr = workloads.soup_scene(a.tris, sky=workloads.hw_sky())
if a.leaf_order:
    order = arr.nodes["shape_index"][arr.nodes["is_leaf"] != 0]
    t2 = np.ascontiguousarray(arr.triangles[order])
    arr = abi.SceneArrays(t2, core.bvh_build(t2), arr.materials, arr.sky)
ctx = core.Context(0)
sc = ctx.scene(arr)
d_img = torch.zeros((a.height, a.width, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
ctx.render_dev(sc, a.width, a.height, 0, 1, 1, 1, abi.PT_RENDER_FINAL, d_img.data_ptr())
rays_total, paths_total = ctx.stats().trace_rays, ctx.stats().paths       # everything this process traces (the counter passes sum over it)
for r in range(a.reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    ctx.render_dev(sc, a.width, a.height, 0, a.spp, a.spp, 1, abi.PT_RENDER_FINAL, d_img.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    st = ctx.stats()
    print(f"{a.spp} spp in {dt*1e3:.1f} ms -> {a.spp/dt:.1f} spp/s | device {st.total_ms:.1f} ms, trace {st.trace_ms:.1f} ms "
          f"({st.trace_rays/st.trace_ms/1e3:.0f} Mrays/s, {st.trace_launches} launches), segments/path {st.segments/st.paths:.3f} | "
          f"shade {st.shade_ms:.1f} sky {st.sky_ms:.1f} accumulate {st.accumulate_ms:.1f} generate {st.generate_ms:.1f} ms", flush=True)
    rays_total += st.trace_rays
    paths_total += st.paths
print(f"profile_totals rays {rays_total} paths {paths_total}", flush=True)
img = d_img.cpu().numpy()
print("image crc", zlib.crc32(np.nan_to_num(img).tobytes()), "finite", float(np.isfinite(img[..., 1]).mean()))
