"""CPU restatement (numpy) of the reference's display transform.  TEST INFRASTRUCTURE ONLY — only
tests/ may import this; the product path is pathtracer_amd/csrc/tonemap.hip.

Follows /root/reference:
  src/renderer.rs:297-309   log-average luminance (sequential f32 sum of ln(0.001 + |Y|)), zones
  src/shader.wgsl:32-44     exposure, extended Reinhard on Y, XYZ -> linear sRGB
  src/renderer.rs:191,204   Bgra8Unorm surface: clamp to [0,1], quantise, no sRGB encoding
Parity unpinned: the reference holds no vectors for this; the constants below are its literals.
"""
import numpy as np

XYZ2SRGB_COLUMNS = np.array([[3.24096994, -1.53738318, -0.49861076],
                             [-0.96924364, 1.8759675, 0.04155506],
                             [0.05563008, -0.20397696, 1.05697151]], dtype=np.float32)   # shader.wgsl:34-38


def tone_stats(xyzw, avg_zone=5.0, white_zone=10.0):
    y = np.abs(xyzw[..., 1].astype(np.float32).ravel())
    logs = np.log(np.float32(0.001) + y).astype(np.float32)
    total = np.float32(0.0)
    # sequential f32 accumulation, as the loop at renderer.rs:299-306 (np.cumsum keeps the order)
    total = np.cumsum(logs, dtype=np.float32)[-1]
    n = np.float32(4 * y.size)                                                 # framebuffer.len()
    avg = np.exp(np.float32(4.0) * total / n).astype(np.float32)               # renderer.rs:307
    mx = np.float32(avg * np.float32(2.0) ** np.float32(white_zone - avg_zone))  # :308
    mid = np.float32(np.float32(0.18) * np.float32(2.0) ** np.float32(avg_zone - 5.0))  # :309
    return np.float32(avg), mx, mid


def tonemap(xyzw, avg_zone=5.0, white_zone=10.0):
    xyzw = np.asarray(xyzw, dtype=np.float32)
    avg, mx, mid = tone_stats(xyzw, avg_zone, white_zone)
    xyz = xyzw[..., :3] * (mid / avg)                                          # shader.wgsl:41
    yy = xyz[..., 1:2]
    xyz = xyz * ((np.float32(1.0) + yy / (mx * mx)) / (np.float32(1.0) + yy))   # :42
    rgb = np.stack([(xyz * XYZ2SRGB_COLUMNS[j]).sum(axis=-1, dtype=np.float32) for j in range(3)], axis=-1)   # :43
    rgba = np.concatenate([rgb, xyzw[..., 3:4]], axis=-1).astype(np.float32)
    q = np.where(np.isnan(rgba), 0.0, np.clip(rgba, 0.0, 1.0))
    u8 = np.floor(q * 255.0 + 0.5).astype(np.uint8)
    bgra8 = u8[..., [2, 1, 0, 3]]
    return rgba, bgra8, (avg, mx, mid)
