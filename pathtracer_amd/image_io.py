"""Image files for looking at renders (the reference writes none, main.rs:113; its window is gone).

PNG from the BGRA8 surface bytes of pt_tonemap (zlib only, no imaging library), PFM for the raw XYZ."""
import struct
import zlib

import numpy as np


def write_png(path, bgra8):
    """8-bit RGBA PNG from the Bgra8Unorm surface bytes ([H, W, 4] uint8, B G R A)."""
    bgra8 = np.ascontiguousarray(bgra8, dtype=np.uint8)
    h, w, _ = bgra8.shape
    rgba = bgra8[..., [2, 1, 0, 3]]
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_pfm(path, xyz):
    """Little-endian colour PFM of the first three channels (rows bottom-up, as the format wants)."""
    xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32)[..., :3])
    h, w, _ = xyz.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(xyz[::-1].astype("<f4").tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data[::-1].copy()
