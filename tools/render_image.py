#!/usr/bin/env python3
"""Render one of the synthetic scenes through the C ABI and write PNG (tone-mapped as the reference's
display would, NaN pixels shown black) + PFM (raw XYZ).
usage: tools/render_image.py box|soup|soup_d65|path/to/scene.toml out_prefix
A .toml argument goes through the library's Scene::load (pt_scene_file_load); resolution and samples then
come from the file unless overridden."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pathtracer_amd import core, image_io, workloads

ap = argparse.ArgumentParser()
ap.add_argument("scene", help="box | soup | soup_d65 | a reference scene .toml")
ap.add_argument("out")
ap.add_argument("--width", type=int, default=None)
ap.add_argument("--height", type=int, default=None)
ap.add_argument("--spp", type=int, default=None)
ap.add_argument("--tris", type=int, default=200_000)
a = ap.parse_args()
cam = {"resolution": (640, 480), "samples": 64, "sensor_width": 36.0, "focal_length": 50.0}
if a.scene.endswith(".toml"):
    loaded = core.load_scene_file(a.scene)
    arr, cam = loaded.arrays, loaded.camera
else:
    arr = {"box": lambda: workloads.box_scene(),
           "soup": lambda: workloads.soup_scene(a.tris, sky=workloads.hw_sky()),
           "soup_d65": lambda: workloads.soup_scene(a.tris, sky=core.sky_d65(1.0))}[a.scene]()
a.width, a.height = a.width or cam["resolution"][0], a.height or cam["resolution"][1]
a.spp = a.spp or cam["samples"]
ctx = core.Context(0)
img = ctx.render(ctx.scene(arr), a.width, a.height, 0, a.spp, a.spp, seed=1,
                 sensor_width=cam["sensor_width"], focal_length=cam["focal_length"])
st = ctx.stats()
rgba, bgra8, stats = ctx.tonemap(img, skip_nonfinite=True)
image_io.write_png(a.out + ".png", bgra8)
image_io.write_pfm(a.out + ".pfm", np.nan_to_num(img))
print(f"{a.scene}: {a.width}x{a.height} {a.spp} spp, {st.segments/st.paths:.2f} segments/path, finite {np.isfinite(img[...,1]).mean():.3f}, "
      f"avg luminance {stats[0]:.4g}, device {st.total_ms:.1f} ms -> {a.out}.png")
