#!/usr/bin/env python3
"""Development aid: what two closest-hit batches in flight are worth.  Two contexts on GPU 0 (each with its own stream and
workspace), one host thread each making blocking pt_trace_closest_dev calls, against one context making the same calls."""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pathtracer_amd import core, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--threads", type=int, default=2)
a = ap.parse_args()

arr = workloads.soup_scene(a.tris)
rays = core.gen_primary_rays(a.width, a.height)
n = rays.size
ctxs, scs, d_rays, d_out = [], [], [], []
for i in range(a.threads):
    c = core.Context(0)
    ctxs.append(c)
    scs.append(c.scene(arr))
    d_rays.append(torch.from_numpy(rays.view(np.float32).reshape(n, 8)).cuda())
    d_out.append(torch.zeros((n, 4), dtype=torch.int32, device="cuda"))
torch.cuda.synchronize()


def work(i, reps):
    for _ in range(reps):
        ctxs[i].trace_dev(scs[i], d_rays[i].data_ptr(), 0, n, d_out[i].data_ptr())


work(0, 5)
t0 = time.perf_counter()
work(0, a.reps)
dt = time.perf_counter() - t0
print(f"one context: {dt / a.reps * 1e3:.3f} ms per batch -> {n * a.reps / dt / 1e6:.1f} Mrays/s", flush=True)
for i in range(a.threads):
    work(i, 3)
th = [threading.Thread(target=work, args=(i, a.reps)) for i in range(a.threads)]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.perf_counter() - t0
print(f"{a.threads} contexts, one thread each: {dt / (a.reps * a.threads) * 1e3:.3f} ms per batch -> {n * a.reps * a.threads / dt / 1e6:.1f} Mrays/s", flush=True)
same = all(bool((d_out[0] == d_out[i]).all()) for i in range(1, a.threads))
print("same hits in every context:", same)
