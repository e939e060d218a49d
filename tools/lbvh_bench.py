#!/usr/bin/env python3
"""Device-side LBVH build against the host SAH build (reference topology): build time, and what the other
topology costs in traversal.  usage: tools/lbvh_bench.py [--tris N] (GPU box)"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pathtracer_amd import abi, core, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--big", action="store_true", help="the 20M-triangle stress scene of config 5 (3840x2160 primaries)")
a = ap.parse_args()
torch.cuda.init()
ctx = core.Context(0)
t0 = time.time()
if a.big:
    tris = core.gen_soup(0x50554F53, 20_000_000, (-4, -4, -12), (4, 4, -4), 0.002, 0.2, 8)
    w, h = 3840, 2160
else:
    tris = workloads.soup_scene(a.tris).triangles
    w, h = 1920, 1080
mats = workloads.soup_scene(8).materials
t1 = time.time()
nodes = core.bvh_build(tris)
host_build_s = time.time() - t1
arr = abi.SceneArrays(tris, nodes, mats, core.sky_d65(1.0))
t1 = time.time(); sah = ctx.scene(arr); sah_create_s = time.time() - t1
t1 = time.time(); lb = ctx.scene(arr, lbvh=True); lbvh_create_s = time.time() - t1
lbvh_device_ms = ctx.stats().total_ms
rays = core.gen_primary_rays(w, h)
n = rays.size
d_rays = torch.from_numpy(rays.view(np.float32).reshape(n, 8)).cuda()
d_hits = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
out = {"triangles": int(tris.size), "host_sah_build_s": host_build_s, "scene_create_with_host_nodes_s": sah_create_s,
       "scene_create_lbvh_s": lbvh_create_s, "lbvh_device_build_ms": lbvh_device_ms,
       "depth": {"sah": sah.info().max_depth, "lbvh": lb.info().max_depth}}
res = {}
for name, sc in (("sah", sah), ("lbvh", lb)):
    for _ in range(3):
        ctx.trace_dev(sc, d_rays.data_ptr(), 0, n, d_hits.data_ptr())
    ms = []
    for _ in range(10):
        ctx.trace_dev(sc, d_rays.data_ptr(), 0, n, d_hits.data_ptr())
        ms.append(ctx.stats().trace_ms)
    res[name] = d_hits.cpu().numpy().copy()
    out[name + "_trace_ms"] = float(np.mean(ms))
    out[name + "_Mrays_per_s"] = n / np.mean(ms) / 1e3
out["hits_bitwise_equal"] = bool((res["sah"] == res["lbvh"]).all())
out["id_mismatches"] = int((res["sah"][:, 0] != res["lbvh"][:, 0]).sum())
print(json.dumps(out))
