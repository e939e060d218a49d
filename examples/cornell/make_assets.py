#!/usr/bin/env python3
"""Writes the example scene's assets: a Cornell-style box (OBJ/MTL, quads), a textured floor (PNG,
nearest-sampled like the reference's Texture::Nearest) and a dissolve mask.  Run once; outputs are committed."""
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def png(path, px):
    px = np.asarray(px, dtype=np.uint8)
    h, w, c = px.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
                           + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


# 8x8 checker floor texture, sRGB-encoded values
y, x = np.mgrid[0:8, 0:8]
check = np.where(((x + y) & 1)[..., None] == 0, np.array([230, 225, 210]), np.array([90, 110, 160])).astype(np.uint8)
png(os.path.join(HERE, "checker.png"), check)
# dissolve mask for the screen: alpha 255 = solid, 0 = passes through (material.rs:54-78, 277-283)
yy, xx = np.mgrid[0:16, 0:16]
mask = np.zeros((16, 16, 4), dtype=np.uint8)
mask[..., :3] = 255
mask[..., 3] = np.where(((xx // 2 + yy // 2) & 1) == 0, 255, 0)
png(os.path.join(HERE, "screen_mask.png"), mask)

V = []
F = []


def quad(a, b, c, d, mat, uv=False):
    base = len(V) + 1
    V.extend([a, b, c, d])
    if uv:
        F.append((mat, "f %d/1 %d/2 %d/3 %d/4" % (base, base + 1, base + 2, base + 3)))
    else:
        F.append((mat, "f %d %d %d %d" % (base, base + 1, base + 2, base + 3)))


def block(x0, x1, y0, y1, z0, z1, mat):
    quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), mat)      # top
    quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), mat)      # front
    quad((x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0), mat)      # back
    quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), mat)      # left
    quad((x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1), mat)      # right


quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), "floor", uv=True)
quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), "white")              # ceiling
quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), "white")          # back wall
quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), "red")
quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), "green")
quad((-0.3, 0.995, -0.3), (0.3, 0.995, -0.3), (0.3, 0.995, 0.3), (-0.3, 0.995, 0.3), "lamp")
block(-0.65, -0.15, -1.0, 0.2, -0.6, -0.1, "white")
block(0.15, 0.6, -1.0, -0.4, -0.1, 0.4, "white")
quad((-0.9, -1.0, 0.55), (-0.1, -1.0, 0.55), (-0.1, -0.2, 0.55), (-0.9, -0.2, 0.55), "screen", uv=True)

with open(os.path.join(HERE, "box.obj"), "w") as f:
    f.write("# Cornell-style box, written by make_assets.py\nmtllib box.mtl\n")
    for v in V:
        f.write("v %g %g %g\n" % v)
    f.write("vt 0 0\nvt 4 0\nvt 4 4\nvt 0 4\n")
    cur = None
    for mat, line in F:
        if mat != cur:
            f.write("usemtl %s\n" % mat)
            cur = mat
        f.write(line + "\n")
with open(os.path.join(HERE, "box.mtl"), "w") as f:
    f.write("newmtl white\nKd 0.73 0.73 0.73\n"
            "newmtl red\nKd 0.65 0.05 0.05\n"
            "newmtl green\nKd 0.12 0.45 0.15\n"
            "newmtl floor\nKd 1 1 1\nmap_Kd checker.png\n"
            "newmtl lamp\nKd 0 0 0\nKe 1.0 0.85 0.6\n"
            "newmtl screen\nKd 0.8 0.7 0.2\nmap_d screen_mask.png\n")
