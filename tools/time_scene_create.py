#!/usr/bin/env python3
"""Development aid: wall time of pt_scene_create for soup-N under PTC_FLATTEN = 0 (device) and 1 (host loop)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pathtracer_amd import abi, core, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
t = time.perf_counter()
tris = workloads.soup20m_triangles() if n == 20_000_000 else workloads.soup_scene(n).triangles
nodes = core.bvh_build(tris)
arr = abi.SceneArrays(tris, nodes, workloads.soup_materials(), workloads.hw_sky())
print(f"soup-{n}: triangles + pt_bvh_build in {time.perf_counter() - t:.2f} s", flush=True)
for flatten in ("0", "1", "0", "1"):
    os.environ["PTC_FLATTEN"] = flatten
    ctx = core.Context(0)
    t = time.perf_counter()
    sc = ctx.scene(arr)
    dt = time.perf_counter() - t
    print(f"PTC_FLATTEN={flatten}: pt_scene_create {dt:.3f} s, {sc.info().n_internal_nodes} internal nodes, depth {sc.info().max_depth}", flush=True)
    sc.close()
    ctx.close()
