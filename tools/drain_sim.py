#!/usr/bin/env python3
"""How much of a 2 M-ray launch is ramp and drain, and what could an ORDER of the rays or a hand-over of subtrees win?  CPU only.

The CPU model of the default traversal kernel (tests/cpp/scoped_model.cpp, the decision code the kernel compiles) gives every
ray's loop trips = the length of its dependent chain of fetches.  A launch is then list scheduling: `lanes` lanes, each takes
the next ray of the order when it is free, a trip costs one time unit (the GPU: 1.2 - 1.5 us with five waves per SIMD).  Printed:
the makespan in trips for
  index order          what the kernel does (tickets in ray order),
  LPT                  longest rays first: the bound of ANY scheme that only reorders rays (it needs an oracle for the lengths),
  predicted            rays ordered by the mean length of an already-traced sparse sample of their 64-ray ticket's neighbourhood
                       (every `--sample`-th ticket traced first, the rest by the nearest sample's mean, longest first),
  split                index order, but a ray longer than `--split-over` trips hands its work out: its chain shrinks to
                       split-over + hand-off (the bound of farming the slowest rays' subtrees out),
and the lower bound total work / lanes.
usage: tools/drain_sim.py [--width 1920 --height 1080] [--lanes 327680] [--cache trips.npy]"""
import argparse
import ctypes as C
import heapq
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pathtracer_amd import abi, core, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--lanes", type=int, default=256 * 5 * 256)
ap.add_argument("--sample", type=int, default=8)
ap.add_argument("--split-over", type=int, default=60)
ap.add_argument("--handoff", type=float, default=2.0, help="hand-off cost in trips")
ap.add_argument("--cache", default="/tmp/drain_sim_trips.npy")
a = ap.parse_args()

if os.path.exists(a.cache):
    trips = np.load(a.cache)
else:
    L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "libscopedmodel.so"))
    L.ptm_trace_scoped_trips.restype = C.c_int
    L.ptm_trace_scoped_trips.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    arr = workloads.soup_scene(a.tris)
    rays = core.gen_primary_rays(a.width, a.height)
    n = rays.size
    trips = np.zeros(n, dtype=np.uint32)
    out = np.zeros(n, dtype=abi.HIT)
    flagged = np.zeros(n, dtype=np.uint8)
    stats = (C.c_uint64 * 8)()
    rc = L.ptm_trace_scoped_trips(abi.ptr(arr.nodes), arr.nodes.size, abi.ptr(arr.triangles), arr.triangles.size, abi.ptr(rays), None, n,
                                  abi.ptr(out), abi.ptr(flagged), stats, abi.ptr(trips))
    assert rc == 0
    np.save(a.cache, trips)
n = trips.size
print(f"{n} rays: trips mean {trips.mean():.1f}, p50 {np.percentile(trips, 50):.0f}, p99 {np.percentile(trips, 99):.0f}, p99.9 {np.percentile(trips, 99.9):.0f}, max {trips.max()}")


def makespan(lengths_in_order, lanes):
    """list scheduling: every lane takes the next ray when it is free"""
    m = lengths_in_order.size
    if m <= lanes:
        return float(lengths_in_order.max())
    free = list(lengths_in_order[:lanes].astype(np.float64))
    heapq.heapify(free)
    for x in lengths_in_order[lanes:]:
        heapq.heapreplace(free, free[0] + float(x))
    return max(free)


lanes = a.lanes
lower = trips.sum() / lanes
base = makespan(trips, lanes)
lpt = makespan(np.sort(trips)[::-1], lanes)
# predicted order: tickets of 64 consecutive rays; every `sample`-th ticket is traced first (its lengths are then known), the other
# tickets are ordered by the mean of the nearest sampled ticket, longest first
T = 64
nt = (n + T - 1) // T
tmean = np.array([trips[i * T:(i + 1) * T].mean() for i in range(nt)])
sampled = np.arange(0, nt, a.sample)
rest = np.setdiff1d(np.arange(nt), sampled)
pred = tmean[(rest // a.sample) * a.sample]                   # the sampled ticket just before it (same image row, a few pixels away)
order_t = np.concatenate([sampled, rest[np.argsort(-pred, kind="stable")]])
order = np.concatenate([np.arange(t * T, min(n, (t + 1) * T)) for t in order_t])
predicted = makespan(trips[order], lanes)
# per-ray prediction from the sampled tickets would be the next step; the ticket mean is what a wave-granular scheduler can use
split = makespan(np.where(trips > a.split_over, a.split_over + a.handoff, trips).astype(np.float64), lanes)
corr = np.corrcoef(pred, tmean[rest])[0, 1]
print(f"lanes {lanes}, rays per lane {n / lanes:.2f}")
print(f"lower bound (work / lanes)      {lower:8.1f} trips")
print(f"index order (the kernel)        {base:8.1f} trips   drain+ramp share {(base - lower) / base:.2f}")
print(f"LPT (oracle order)              {lpt:8.1f} trips   {100 * (1 - lpt / base):5.1f} % shorter")
print(f"predicted order (1/{a.sample} sampled)  {predicted:8.1f} trips   {100 * (1 - predicted / base):5.1f} % shorter   (ticket-mean correlation with its sampled neighbour {corr:.2f})")
print(f"split over {a.split_over} trips (+{a.handoff})     {split:8.1f} trips   {100 * (1 - split / base):5.1f} % shorter (chain only: the farmed-out work is assumed to cost no lane time)")
