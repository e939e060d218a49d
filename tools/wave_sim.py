#!/usr/bin/env python3
"""Where the lanes go: replays the per-trip logs of the CPU model of the default traversal kernel (tests/cpp/scoped_model.cpp,
the same decision code the kernel compiles) through a 64-lane wave with the kernel's refill rule, and counts, per block of
the loop (record step, leaf test, pop iterations), how many lanes are active when the wave executes it.  CPU only.
usage: tools/wave_sim.py [--tris N] [--first RAY] [--rays N] [--refill 16] [--bounce]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pathtracer_amd import abi, core, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--first", type=int, default=900_000)
ap.add_argument("--rays", type=int, default=64 * 1024)
ap.add_argument("--refill", type=int, default=16)
ap.add_argument("--bounce", action="store_true")
a = ap.parse_args()
COST = {"record": 115, "leaf": 69, "pop": 15}        # VALU per execution (profiles/issue_roof.json: wide step + pushes, leaf, pop iteration)

L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "libscopedmodel.so"))
L.ptm_trace_scoped_log.restype = C.c_uint64
L.ptm_trace_scoped_log.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
arr = workloads.soup_scene(a.tris)
rays = core.gen_primary_rays(1920, 1080)
start = None
if a.bounce:
    from tests.golden import fixture_inputs as fi
    from tests.oracle import Oracle
    o = Oracle()
    sub = np.ascontiguousarray(rays[a.first:a.first + a.rays])
    rays_b, start = fi.bounce_rays(sub, o.trace(o.scene(arr), sub))
    r = np.ascontiguousarray(rays_b)
else:
    r = np.ascontiguousarray(rays[a.first:a.first + a.rays])
n = L.ptm_trace_scoped_log(abi.ptr(arr.nodes), arr.nodes.size, abi.ptr(arr.triangles), arr.triangles.size, abi.ptr(r), abi.ptr(start), r.size, None, 0)
log = np.zeros(n, dtype=np.uint8)
L.ptm_trace_scoped_log(None, 0, None, 0, None, None, 0, abi.ptr(log), n)
ends = np.flatnonzero(log == 0xFF)
begins = np.concatenate([[0], ends[:-1] + 1])
trips = [log[b:e] for b, e in zip(begins, ends)]
print(f"{len(trips)} rays, {np.mean([t.size for t in trips]):.1f} trips per ray")

# event sequence of every ray: R (record step), L (leaf test), P (one pop), in the order the kernel's loop takes them
EV_R, EV_L, EV_P = 0, 1, 2
events = []
for t in trips:
    ev = []
    for b in t:
        b = int(b)
        if b & 1: ev.append(EV_R)
        if b & 2: ev.append(EV_L)
        ev.extend([EV_P] * (b >> 2))
    events.append(np.array(ev, dtype=np.uint8))


def simulate(pop_cap, leaf_min, refill=a.refill, overhead=12):
    """pop_cap: pop iterations per wave trip (0 = until every lane has found an entry); leaf_min: the leaf block runs when at
    least this many lanes wait at a leaf, or when no lane has a record to visit."""
    wave = {"record": 0, "leaf": 0, "pop": 0, "loop": 0}
    lane = {"record": 0, "leaf": 0, "pop": 0}
    nxt, ntrips = 0, 0
    ray = [None] * 64
    pos = [0] * 64
    while True:
        idle = [i for i in range(64) if ray[i] is None]
        if nxt < len(events) and (len(idle) >= refill or len(idle) == 64):
            for i in idle:
                if nxt >= len(events):
                    break
                ray[i], pos[i] = events[nxt], 0
                nxt += 1
                if ray[i].size == 0:
                    ray[i] = None
        if nxt >= len(events):
            break
        ntrips += 1
        wave["loop"] += overhead

        def nextev(i):
            return int(ray[i][pos[i]]) if ray[i] is not None else -1

        def advance(i):
            pos[i] += 1
            if pos[i] >= ray[i].size:
                ray[i] = None
        rl = [i for i in range(64) if nextev(i) == EV_R]
        if rl:
            wave["record"] += COST["record"]; lane["record"] += COST["record"] * len(rl)
            for i in rl:
                advance(i)
        ll = [i for i in range(64) if nextev(i) == EV_L]
        if ll and (len(ll) >= leaf_min or not rl):
            wave["leaf"] += COST["leaf"]; lane["leaf"] += COST["leaf"] * len(ll)
            for i in ll:
                advance(i)
        it = 0
        while True:
            pl = [i for i in range(64) if nextev(i) == EV_P]
            if not pl or (pop_cap and it >= pop_cap):
                break
            wave["pop"] += COST["pop"]; lane["pop"] += COST["pop"] * len(pl)
            for i in pl:
                advance(i)
            it += 1
    tw = sum(wave.values())
    return tw / nxt, ntrips, {k: (wave[k] / tw, lane[k] / max(wave[k], 1)) for k in lane}


print("pop_cap leaf_min | wave VALU instructions per ray | wave trips | share / active lanes: record, leaf, pop")
for pop_cap in (0, 1, 2, 3):
    for leaf_min in (1, 8, 16, 24, 32):
        per_ray, ntrips, d = simulate(pop_cap, leaf_min)
        print(f"{pop_cap:7d} {leaf_min:8d} | {per_ray:7.1f} | {ntrips:6d} | " + ", ".join(f"{k} {d[k][0]*100:4.1f}% {d[k][1]:4.1f}" for k in d), flush=True)
