#!/usr/bin/env python3
"""Writes the example's assets: three UV spheres with per-vertex normals (smooth shading through the reference's
normal interpolation, material.rs:279-282) on a textured ground quad, lit by a D65 sky.  Outputs are committed."""
import math
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def png(path, px):
    px = np.asarray(px, dtype=np.uint8)
    h, w, c = px.shape
    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                           + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


y, x = np.mgrid[0:16, 0:16]
tiles = np.where((((x // 2) + (y // 2)) & 1)[..., None] == 0, np.array([200, 190, 170]), np.array([120, 100, 80])).astype(np.uint8)
png(os.path.join(HERE, "tiles.png"), tiles)

lines = ["# three smooth spheres on a ground quad, written by make_assets.py", "mtllib scene.mtl"]
nv = 0


def sphere(cx, cy, cz, r, mat, nu=48, nvv=24):
    global nv
    base = nv
    for j in range(nvv + 1):
        th = math.pi * j / nvv
        for i in range(nu):
            ph = 2 * math.pi * i / nu
            n = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
            lines.append("v %.6f %.6f %.6f" % (cx + r * n[0], cy + r * n[1], cz + r * n[2]))
            lines.append("vn %.6f %.6f %.6f" % n)
    lines.append("usemtl " + mat)
    for j in range(nvv):
        for i in range(nu):
            a = base + j * nu + i + 1
            b = base + j * nu + (i + 1) % nu + 1
            c = base + (j + 1) * nu + (i + 1) % nu + 1
            d = base + (j + 1) * nu + i + 1
            if j == 0:
                lines.append("f %d//%d %d//%d %d//%d" % (a, a, c, c, d, d))
            elif j == nvv - 1:
                lines.append("f %d//%d %d//%d %d//%d" % (a, a, b, b, d, d))
            else:
                lines.append("f %d//%d %d//%d %d//%d %d//%d" % (a, a, b, b, c, c, d, d))
    nv += (nvv + 1) * nu


sphere(-1.3, 0.6, 0.0, 0.6, "red")
sphere(0.0, 0.8, -0.4, 0.8, "white")
sphere(1.4, 0.5, 0.3, 0.5, "blue")
g = nv
for p in ((-6, 0, 6), (6, 0, 6), (6, 0, -6), (-6, 0, -6)):
    lines.append("v %g %g %g" % p)
lines += ["vt 0 0", "vt 6 0", "vt 6 6", "vt 0 6", "usemtl ground",
          "f %d/1 %d/2 %d/3 %d/4" % (g + 1, g + 2, g + 3, g + 4)]
open(os.path.join(HERE, "scene.obj"), "w").write("\n".join(lines) + "\n")
open(os.path.join(HERE, "scene.mtl"), "w").write(
    "newmtl red\nKd 0.75 0.12 0.1\nnewmtl white\nKd 0.8 0.8 0.8\nnewmtl blue\nKd 0.1 0.2 0.7\n"
    "newmtl ground\nKd 1 1 1\nmap_Kd tiles.png\n")
