#!/usr/bin/env python3
"""Development aid: how fast the default traversal is on the rays of the k-th bounce (the integrator's later iterations:
survivors of k scatterings, in the order the queue would hold them), and what sorting them by origin would buy."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pathtracer_amd import abi, core, workloads
from tests.golden import fixture_inputs as fi

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=4)
ap.add_argument("--spp", type=int, default=4, help="copies of every pixel's ray (stand-ins for the samples of a pixel), side by side")
a = ap.parse_args()
arr = workloads.soup_scene(1_000_000)
ctx = core.Context(0)
sc = ctx.scene(arr)
rays = np.repeat(core.gen_primary_rays(1920, 1080), a.spp)
start = None


def spread(v):
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    return (v | (v << 2)) & 0x09249249


def timed(rays, start):
    n = rays.size
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(n, 8)).cuda()
    d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    d_start = None if start is None else torch.from_numpy(start.view(np.int32)).cuda()
    ms = []
    for _ in range(5):
        ctx.trace_dev(sc, d_rays.data_ptr(), 0 if d_start is None else d_start.data_ptr(), n, d_out.data_ptr())
        ms.append(ctx.stats().trace_ms)
    return float(np.mean(ms[1:])), d_out.cpu().numpy().view(abi.HIT).reshape(-1)


for depth in range(a.depth + 1):
    ms, hits = timed(rays, start)
    line = f"bounce {depth}: {rays.size} rays, queue order {rays.size / ms / 1e3:.0f} Mrays/s"
    if depth >= 1:
        o = rays["o"]
        q = np.clip(((o - o.min(axis=0)) / (np.ptp(o, axis=0) + 1e-20) * 1023).astype(np.uint64), 0, 1023)
        order = np.argsort((spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2]), kind="stable")
        ms2, _ = timed(np.ascontiguousarray(rays[order]), np.ascontiguousarray(start[order]))
        line += f", sorted by a 30-bit Morton key of the origin {rays.size / ms2 / 1e3:.0f} Mrays/s"
    print(line, flush=True)
    keep = hits["tri"] != abi.PT_NO_TRIANGLE
    # the integrator's survival: a hit, then Russian roulette - here a fixed fraction, in queue order
    keep &= (np.arange(rays.size) * 2654435761 % 1000) < (1000 if depth < 2 else 700)
    nrays, nstart = fi.bounce_rays(rays, hits)
    # decorrelate the directions between depths (bounce_rays hashes the index)
    rays, start = np.ascontiguousarray(nrays[keep]), np.ascontiguousarray(nstart[keep])
