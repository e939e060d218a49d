#!/usr/bin/env python3
"""Render one of the synthetic scenes through the C ABI and write PNG (tone-mapped as the reference's
display would, NaN pixels shown black) + PFM (raw XYZ).  usage: tools/render_image.py box|soup out_prefix"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pathtracer_amd import core, image_io, workloads

ap = argparse.ArgumentParser()
ap.add_argument("scene", choices=["box", "soup", "soup_d65"])
ap.add_argument("out")
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--tris", type=int, default=200_000)
a = ap.parse_args()
arr = {"box": lambda: workloads.box_scene(),
       "soup": lambda: workloads.soup_scene(a.tris, sky=workloads.hw_sky()),
       "soup_d65": lambda: workloads.soup_scene(a.tris, sky=core.sky_d65(1.0))}[a.scene]()
ctx = core.Context(0)
img = ctx.render(ctx.scene(arr), a.width, a.height, 0, a.spp, a.spp, seed=1)
st = ctx.stats()
rgba, bgra8, stats = ctx.tonemap(img, skip_nonfinite=True)
image_io.write_png(a.out + ".png", bgra8)
image_io.write_pfm(a.out + ".pfm", np.nan_to_num(img))
print(f"{a.scene}: {a.width}x{a.height} {a.spp} spp, {st.segments/st.paths:.2f} segments/path, finite {np.isfinite(img[...,1]).mean():.3f}, "
      f"avg luminance {stats[0]:.4g}, device {st.total_ms:.1f} ms -> {a.out}.png")
