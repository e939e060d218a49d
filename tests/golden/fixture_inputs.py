"""Deterministic INPUTS of the golden fixtures (exact integer / binary32 arithmetic only)."""
import zlib

import numpy as np

from pathtracer_amd import abi, core


def primary_subset(width, height, n):
    """n pixel-centre rays spread evenly over the width x height grid (Camera::generate_ray)."""
    rays = core.gen_primary_rays(width, height)
    idx = (np.arange(n, dtype=np.uint64) * np.uint64(rays.size)) // np.uint64(n)
    return rays[idx.astype(np.int64)].copy()


def _hash_u32(x):
    """lowbias32 (public-domain integer hash) on uint32 arrays."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x.astype(np.uint64) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def _unit_float(h):
    """[0,1) from the top 24 bits: exact in binary32."""
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def bounce_rays(rays, hits):
    """Secondary rays as the integrator makes them (scene.rs:355-356): origin = o + d*t of ray i's
    hit (rays that missed restart from the camera ray's origin with start = none), direction in
    [-1,1)^3 scaled by a factor in [1/64, 2) — un-normalised, like `normal + UnitSphere`."""
    n = rays.size
    i = np.arange(n, dtype=np.uint32)
    d = np.stack([_unit_float(_hash_u32(i * np.uint32(4) + np.uint32(k))) for k in range(3)], axis=1)
    d = d * np.float32(2.0) - np.float32(1.0)
    scale = np.float32(1.0 / 64.0) + _unit_float(_hash_u32(i * np.uint32(4) + np.uint32(3))) * np.float32(2.0 - 1.0 / 64.0)
    out = np.zeros(n, dtype=abi.RAY)
    hit = hits["tri"] != abi.PT_NO_TRIANGLE
    t = np.where(hit, hits["t"], np.float32(0.0)).astype(np.float32)
    out["o"] = rays["o"] + rays["d"] * t[:, None]
    out["d"] = d * scale[:, None]
    start = np.where(hit, hits["tri"], abi.PT_NO_TRIANGLE).astype(np.uint32)
    return out, start


def sky_probe():
    """48 un-normalised directions (upper and lower hemisphere, a few aimed at the config-3 sun) and
    3 wavelength sets."""
    i = np.arange(48, dtype=np.uint32)
    d = np.stack([_unit_float(_hash_u32(i * np.uint32(8) + np.uint32(k + 100))) for k in range(3)], axis=1)
    d = (d * np.float32(2.0) - np.float32(1.0)) * np.float32(1.5)
    sun = np.float32([0.3, 0.8, 0.5])
    sun = sun / np.sqrt((sun * sun).sum(dtype=np.float32))
    d[:4] = sun * np.float32([[1.0], [0.5], [1.0], [1.0]])
    d[2] += np.float32([0.002, 0.0, 0.0])          # just inside the 0.255 degree disc
    d[3] += np.float32([0.0, 0.0, 0.02])           # just outside
    for k in (2, 3):                               # unit length, so that gamma is the true angle
        d[k] = d[k] / np.sqrt((d[k] * d[k]).sum(dtype=np.float32))
    wls = [np.float32(360.0) + (np.float32(h) + np.float32(470.0) * (np.arange(8, dtype=np.float32) / np.float32(8.0)))
           for h in (0.0, 17.25, 58.0)]
    return d.astype(np.float32), wls


def checksum(hits):
    """CRC32 over the (tri, t, u, v) records — a checksum the full-size GPU test compares."""
    return zlib.crc32(np.ascontiguousarray(hits).tobytes()) & 0xFFFFFFFF


def image_checksum(img):
    """CRC32 of an XYZW float image's bits with every NaN canonicalised (NaN payloads and signs are not portable)."""
    a = np.ascontiguousarray(img, dtype=np.float32).copy()
    u = a.view(np.uint32)
    u[np.isnan(a)] = np.uint32(0x7FC00000)
    return zlib.crc32(u.tobytes()) & 0xFFFFFFFF
