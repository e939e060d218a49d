#!/usr/bin/env python3
"""CPU sweep of the scoped-traversal model (tests/cpp/scoped_model.cpp, which shares trace_step.h with the kernel) against the
oracle's traverse_recursive over the parity sweep's scenes: prints one JSON line.  usage: tools/model_sweep.py FIRST COUNT [PROCS]"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(seed):
    import ctypes as C
    import numpy as np
    from pathtracer_amd import abi, core
    from tests import fuzz_scenes
    from tests.oracle import Oracle
    L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "libscopedmodel.so"))
    L.ptm_trace_scoped.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    o = Oracle()
    o.nthreads = 1
    rng = np.random.default_rng(seed)
    tris, mats, sky, scale, images = fuzz_scenes.random_scene(rng)
    try:
        nodes = core.bvh_build(tris)
    except core.PtError:
        return {"seed": seed, "rays": 0, "bad": 0, "flagged": 0, "box": 0}
    arr = abi.SceneArrays(tris, nodes, mats, sky, images)
    osc = o.scene(arr)
    rays = fuzz_scenes.random_rays(rng, 20000, scale)
    want = o.trace(osc, rays)
    start = np.where(rng.random(rays.size) < 0.3, want["tri"], abi.PT_NO_TRIANGLE).astype(np.uint32)
    res = {"seed": seed, "rays": 0, "bad": 0, "flagged": 0, "box": 0}
    for st in (None, start):
        w = o.trace(osc, rays, st)
        out = np.zeros(rays.size, dtype=abi.HIT)
        fl = np.zeros(rays.size, dtype=np.uint8)
        stats = (C.c_uint64 * 8)()
        L.ptm_trace_scoped(abi.ptr(nodes), nodes.size, abi.ptr(tris), tris.size, abi.ptr(rays), abi.ptr(st), rays.size,
                           abi.ptr(out), abi.ptr(fl), stats)
        ok = fl == 0
        res["rays"] += rays.size
        res["flagged"] += int((~ok).sum())
        res["box"] += int(stats[0])
        res["bad"] += int((ok & (out.view(np.uint8).reshape(rays.size, -1) != w.view(np.uint8).reshape(rays.size, -1)).any(axis=1)).sum())
    return res


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    tot = {"scenes": 0, "rays": 0, "bad": 0, "flagged": 0, "box": 0, "bad_seeds": []}
    with ProcessPoolExecutor(procs) as ex:
        for r in ex.map(one, range(first, first + count), chunksize=4):
            tot["scenes"] += 1
            for k in ("rays", "bad", "flagged", "box"):
                tot[k] += r[k]
            if r["bad"] or r["box"]:
                tot["bad_seeds"].append(r["seed"])
    print(json.dumps(tot))
