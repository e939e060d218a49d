"""ctypes / numpy mirrors of include/ptcore.h (plain data only).

numpy structured dtypes let the harness build million-element triangle / node / ray arrays
without Python loops; each dtype's itemsize is asserted against the C layout.
"""
import ctypes as C

import numpy as np

PT_OK = 0
PT_ERR_INVALID = 1
PT_ERR_HIP = 2
PT_ERR_NOMEM = 3
PT_ERR_UNSUPPORTED = 4
PT_ERR_BVH_DEGENERATE = 5
PT_NO_TRIANGLE = 0xFFFFFFFF

PT_TRI_HAS_UV = 1
PT_TRI_HAS_NORMALS = 2
PT_TEX_FLAT = 0
PT_TEX_NEAREST = 1
PT_TEX_BILINEAR = 2
PT_EMIT_NULL = 0
PT_EMIT_ABSORBED_D65 = 1
PT_SKY_D65 = 0
PT_SKY_HOSEK_WILKIE = 1
PT_TRAVERSE_MIN_T = 0
PT_TRAVERSE_EXACT = 1
PT_TRAVERSE_REPLAY = 2
PT_TRAVERSE_DEFAULT = PT_TRAVERSE_EXACT
PT_NOALLOC_MAX_RAYS = 4194304
PT_TRACE_SLOTS = 4
PT_RENDER_FINAL = 0
PT_RENDER_SUM = 1
PT_TONEMAP_SKIP_NONFINITE = 1
PT_SCENE_BUILD_LBVH = 1
PT_RENDER_NORMALIZE_SKY_DIRECTION = 0x100

f4, u4 = np.float32, np.uint32

AABB = np.dtype([("min", f4, 3), ("max", f4, 3)])
BVH_NODE = np.dtype([
    ("is_leaf", u4), ("shape_index", u4), ("child_l_index", u4), ("child_r_index", u4),
    ("child_l_aabb", AABB), ("child_r_aabb", AABB)])
TRIANGLE = np.dtype([
    ("a", f4, 3), ("b", f4, 3), ("c", f4, 3), ("uv", f4, (3, 2)), ("vertex_normals", f4, (3, 3)),
    ("material_index", u4), ("flags", u4)])
RAY = np.dtype([("o", f4, 3), ("pad0", f4), ("d", f4, 3), ("pad1", f4)])
HIT = np.dtype([("tri", u4), ("t", f4), ("u", f4), ("v", f4)])
TEXTURE = np.dtype([("kind", u4), ("value", f4, 3), ("image", u4)])
MATERIAL = np.dtype([
    ("diffuse", TEXTURE), ("emit_kind", u4), ("absorption", TEXTURE), ("base_dissolve", f4),
    ("dissolve", TEXTURE)])

assert AABB.itemsize == 24 and BVH_NODE.itemsize == 64 and TRIANGLE.itemsize == 104
assert RAY.itemsize == 32 and HIT.itemsize == 16 and TEXTURE.itemsize == 20 and MATERIAL.itemsize == 68


class Sky(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("power", C.c_float), ("configs", (C.c_float * 9) * 11),
                ("radiances", C.c_float * 11), ("turbidity", C.c_float), ("solar_radius", C.c_float),
                ("zenith", C.c_float * 3), ("sun", C.c_float * 3)]


class Camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sensor_width", C.c_float),
                ("focal_length", C.c_float)]


class Image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32),
                ("texels", C.POINTER(C.c_float))]


class SceneDesc(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("n_triangles", C.c_uint32),
                ("nodes", C.c_void_p), ("n_nodes", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("images", C.POINTER(Image)), ("n_images", C.c_uint32),
                ("sky", Sky)]


class SceneFileInfo(C.Structure):
    _fields_ = [("desc", SceneDesc), ("camera", Camera), ("samples", C.c_uint32), ("camera_transform", C.c_float * 16)]


class Stats(C.Structure):
    _fields_ = [("trace_ms", C.c_double), ("trace_rays", C.c_uint64), ("trace_launches", C.c_uint32),
                ("total_ms", C.c_double), ("paths", C.c_uint64), ("segments", C.c_uint64),
                ("shade_ms", C.c_double), ("sky_ms", C.c_double), ("accumulate_ms", C.c_double), ("generate_ms", C.c_double)]


class SceneInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_internal_nodes", C.c_uint32), ("max_depth", C.c_uint32),
                ("device_bytes", C.c_uint64)]


assert C.sizeof(Sky) == 480 and C.sizeof(Camera) == 16


def ptr(arr):
    """void* of a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return C.c_void_p(arr.ctypes.data)


def default_material():
    """Material::default() (material.rs:300-309) as one MATERIAL record."""
    m = np.zeros((), dtype=MATERIAL)
    m["diffuse"]["value"] = 0.75
    m["base_dissolve"] = 1.0
    m["dissolve"]["value"][0] = 1.0
    return m


def flat_material(rgb, emit_rgb=None, base_dissolve=1.0):
    m = default_material()
    m["diffuse"]["value"] = rgb
    if emit_rgb is not None:
        m["emit_kind"] = PT_EMIT_ABSORBED_D65
        m["absorption"]["value"] = emit_rgb
    m["base_dissolve"] = base_dissolve
    return m


class SceneArrays:
    """Keeps the numpy arrays of a scene description alive and builds the C struct over them."""

    def __init__(self, triangles, nodes, materials, sky, images=()):
        self.triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE)
        self.nodes = None if nodes is None else np.ascontiguousarray(nodes, dtype=BVH_NODE)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL)
        self.sky = sky
        self.images = [(np.ascontiguousarray(t, dtype=np.float32)) for t in images]

    def desc(self):
        d = SceneDesc()
        d.triangles = self.triangles.ctypes.data if self.triangles.size else None
        d.n_triangles = self.triangles.size
        if self.nodes is not None and self.nodes.size:
            d.nodes = self.nodes.ctypes.data
            d.n_nodes = self.nodes.size
        else:
            d.nodes = None
            d.n_nodes = 0
        d.materials = self.materials.ctypes.data
        d.n_materials = self.materials.size
        if self.images:
            arr = (Image * len(self.images))()
            for i, t in enumerate(self.images):
                h, w = t.shape[0], t.shape[1]
                ch = 1 if t.ndim == 2 else t.shape[2]
                arr[i] = Image(w, h, ch, t.ctypes.data_as(C.POINTER(C.c_float)))
            self._img_arr = arr
            d.images = arr
            d.n_images = len(self.images)
        else:
            d.images = None
            d.n_images = 0
        d.sky = self.sky
        return d
