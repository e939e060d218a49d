#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ with the CPU oracle.

    python tests/golden/make_fixtures.py

PROVENANCE.  The reference (Rust) cannot be built or run in this image and holds no vectors for
BVH build, traversal, materials, sky or the integrator, so these files are outputs
of THIS repository's line-by-line C++ restatement (oracle/pt_oracle.cpp), whose Triangle / camera /
colour code is pinned by the reference's own KATs in tests/test_oracle_kat.py.  For everything
else they are regression pins of the restatement ("parity unpinned" against the reference binary).

Inputs are regenerated, not stored: scenes come from pt_gen_soup / pt_bvh_build, rays from
fixture_inputs.py (exact integer / binary32 arithmetic only, so every platform rebuilds the same
bits).  Stored: expected hits, visit counts, small XYZ images, sky radiances.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from pathtracer_amd import abi, core, workloads  # noqa: E402
from tests.golden import fixture_inputs as fi  # noqa: E402
from tests.oracle import Oracle  # noqa: E402


def main():
    o = Oracle()
    # ---- config 2: soup-1M, 65 536 primary rays of the 1080p grid + 65 536 bounce rays -----------
    arr = workloads.soup_scene(1_000_000)
    sc = o.scene(arr)
    prim = fi.primary_subset(1920, 1080, 65536)
    hp = o.trace(sc, prim)
    brays, bstart = fi.bounce_rays(prim, hp)
    hb = o.trace(sc, brays, bstart)
    np.savez_compressed(os.path.join(HERE, "soup1m_hits_65536.npz"), primary=hp, bounce=hb, bounce_start=bstart)
    print("soup-1M: primary hit rate %.3f, bounce hit rate %.3f" % (
        (hp["tri"] != abi.PT_NO_TRIANGLE).mean(), (hb["tri"] != abi.PT_NO_TRIANGLE).mean()))

    # visit counts over the full 1080p batch: the n_node / n_tri of the roofline
    full = core.gen_primary_rays(1920, 1080)
    hits, cnt = o.trace(sc, full, count=True)
    _, gcnt = o.trace_globalbest_stats(sc, full[::16])
    counts = {
        "tris": 1_000_000, "width": 1920, "height": 1080, "rays": int(full.size),
        "n_node": cnt[0] / full.size, "n_tri": cnt[1] / full.size,
        "rule": "reference recursion (bvh.rs:250-338): far child entered iff near subtree's hit is farther; no global pruning",
        "global_best_n_node": gcnt[0] / full[::16].size, "global_best_n_tri": gcnt[1] / full[::16].size,
        "global_best_note": "what the FAST device traversal visits (every 16th ray)",
        "hit_fraction": float((hits["tri"] != abi.PT_NO_TRIANGLE).mean()),
        "hits_checksum": fi.checksum(hits),
    }
    json.dump(counts, open(os.path.join(HERE, "soup1m_primary_counts.json"), "w"), indent=1)
    print(counts)
    sc.close()

    # ---- small images ------------------------------------------------------------------------------
    images = {}
    box = workloads.box_scene()
    images["box_64x48_4spp_seed3"] = o.render(o.scene(box), 64, 48, 0, 4, 4, seed=3)
    soup = workloads.soup_scene(20_000, sky=workloads.hw_sky())
    img, rc = o.render(o.scene(soup), 80, 45, 0, 4, 4, seed=1, count=True)
    images["soup20k_hw_80x45_4spp_seed1"] = img
    images["soup20k_hw_80x45_sum_2to4_seed1"] = o.render(o.scene(soup), 80, 45, 2, 2, 4, seed=1, mode=abi.PT_RENDER_SUM)
    np.savez_compressed(os.path.join(HERE, "images.npz"), **images)
    json.dump({"soup20k_hw_80x45_4spp_seed1": {"paths": rc[0], "segments": rc[1], "nodes": rc[2], "tris": rc[3]}},
              open(os.path.join(HERE, "images_counts.json"), "w"), indent=1)

    # ---- sky radiances: evaluate_sky on fixed directions / wavelengths -------------------------------
    sky_sc = o.scene(abi.SceneArrays(np.zeros(0, dtype=abi.TRIANGLE), None,
                                     np.array([abi.default_material()], dtype=abi.MATERIAL), workloads.hw_sky()))
    dirs, wls = fi.sky_probe()
    rad = np.stack([o.evaluate_sky(sky_sc, d, wls[i % len(wls)]) for i, d in enumerate(dirs)])
    np.savez_compressed(os.path.join(HERE, "sky_probe.npz"), radiance=rad)
    print("sky probe: finite %.2f, max %.3g" % (np.isfinite(rad).mean(), np.nanmax(rad)))


if __name__ == "__main__":
    main()
