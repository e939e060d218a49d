"""ctypes binding of libptcore.so — the Python-side stand-in for the `ptcore-sys` FFI crate.

Everything that computes goes through the C ABI of include/ptcore.h; there is no Python or CPU
fallback: if the HIP library is missing, loading fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import abi

# PTCORE_LIB: development override used to A/B experimental builds of the same library
_LIB_PATH = os.environ.get("PTCORE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libptcore.so")
_lib = None


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ptcore error {code}: {msg}")
        self.code = code


def lib():
    """The loaded libptcore.so (built in-tree by `make -C pathtracer_amd/csrc` / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
            "(there is no CPU fallback for the render core)")
    L = C.CDLL(_LIB_PATH)
    vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int
    sig = {
        "pt_abi_version": (i32, []),
        "pt_init": (i32, [C.POINTER(i32), i32, C.POINTER(vp)]),
        "pt_shutdown": (None, [vp]),
        "pt_last_error": (C.c_char_p, [vp]),
        "pt_bvh_build": (i32, [vp, u32, vp, C.POINTER(u32)]),
        "pt_sky_hosek_wilkie": (i32, [f32, f32, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(abi.Sky)]),
        "pt_sky_d65": (None, [f32, C.POINTER(abi.Sky)]),
        "pt_material_default": (None, [vp]),
        "pt_scene_create": (i32, [vp, C.POINTER(abi.SceneDesc), C.POINTER(vp)]),
        "pt_scene_create_ex": (i32, [vp, C.POINTER(abi.SceneDesc), u32, C.POINTER(vp)]),
        "pt_scene_destroy": (None, [vp, vp]),
        "pt_trace_closest": (i32, [vp, vp, vp, vp, u32, vp]),
        "pt_trace_closest_dev": (i32, [vp, vp, vp, vp, u32, vp]),
        "pt_trace_closest_dev_noalloc": (i32, [vp, vp, vp, vp, u32, vp]),
        "pt_trace_closest_dev_submit": (i32, [vp, vp, vp, vp, u32, vp, u32]),
        "pt_trace_closest_dev_wait": (i32, [vp, u32]),
        "pt_set_traversal": (i32, [vp, u32]),
        "pt_render": (i32, [vp, vp, C.POINTER(abi.Camera), u32, u32, u32, u64, u32, vp]),
        "pt_render_dev": (i32, [vp, vp, C.POINTER(abi.Camera), u32, u32, u32, u64, u32, vp]),
        "pt_tonemap": (i32, [vp, vp, u32, u32, f32, f32, u32, vp, vp, C.POINTER(f32)]),
        "pt_tonemap_dev": (i32, [vp, vp, u32, u32, f32, f32, u32, vp, vp, C.POINTER(f32)]),
        "pt_get_stats": (i32, [vp, C.POINTER(abi.Stats)]),
        "pt_scene_get_info": (i32, [vp, C.POINTER(abi.SceneInfo)]),
        "pt_scene_get_bvh": (i32, [vp, vp, vp, u32, C.POINTER(u32)]),
        "pt_gen_soup": (i32, [u64, u32, C.POINTER(f32), C.POINTER(f32), f32, f32, u32, vp]),
        "pt_gen_primary_rays": (i32, [C.POINTER(abi.Camera), vp]),
        "pt_host_alloc": (i32, [C.c_size_t, C.POINTER(vp)]),
        "pt_host_free": (None, [vp]),
        "pt_scene_file_load": (i32, [C.c_char_p, C.POINTER(vp)]),
        "pt_scene_file_get": (i32, [vp, C.POINTER(abi.SceneFileInfo)]),
        "pt_scene_file_free": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTS = [
    "pt_abi_version", "pt_init", "pt_shutdown", "pt_last_error", "pt_bvh_build", "pt_sky_hosek_wilkie",
    "pt_sky_d65", "pt_material_default", "pt_scene_create", "pt_scene_destroy", "pt_trace_closest",
    "pt_trace_closest_dev", "pt_set_traversal", "pt_render", "pt_render_dev", "pt_get_stats",
    "pt_scene_get_info", "pt_gen_soup", "pt_gen_primary_rays", "pt_tonemap", "pt_tonemap_dev",
    "pt_scene_file_load", "pt_scene_file_get", "pt_scene_file_free", "pt_scene_create_ex",
    "pt_host_alloc", "pt_host_free", "pt_scene_get_bvh", "pt_trace_closest_dev_noalloc",
    "pt_trace_closest_dev_submit", "pt_trace_closest_dev_wait",
]


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


# ---- host-side mirrors (no GPU needed) -------------------------------------------------------

def bvh_build(triangles):
    """BVH::build (bvh.rs:346-356): reference-order node vector for `triangles`."""
    tris = np.ascontiguousarray(triangles, dtype=abi.TRIANGLE)
    n = tris.size
    nodes = np.zeros(max(2 * n - 1, 0), dtype=abi.BVH_NODE)
    cnt = C.c_uint32(0)
    rc = lib().pt_bvh_build(abi.ptr(tris) if n else None, n, abi.ptr(nodes) if n else None, C.byref(cnt))
    if rc != abi.PT_OK:
        raise PtError(rc, "pt_bvh_build failed")
    return nodes[:cnt.value]


def sky_d65(power):
    s = abi.Sky()
    lib().pt_sky_d65(float(power), C.byref(s))
    return s


def sky_hosek_wilkie(turbidity, albedo, zenith, sun, camera_transform=None):
    s = abi.Sky()
    m = None
    if camera_transform is not None:
        flat = np.asarray(camera_transform, dtype=np.float32).reshape(16)   # column-major
        m = (C.c_float * 16)(*flat.tolist())
    rc = lib().pt_sky_hosek_wilkie(float(turbidity), float(albedo), _f3(zenith), _f3(sun), m, C.byref(s))
    if rc != abi.PT_OK:
        raise PtError(rc, "pt_sky_hosek_wilkie: invalid sky (elevation <= 0 or turbidity outside [1,10])")
    return s


def gen_soup(seed, n, lo, hi, edge_scale, log_scale_hi=0.0, n_material_cycle=8):
    out = np.zeros(n, dtype=abi.TRIANGLE)
    rc = lib().pt_gen_soup(int(seed), int(n), _f3(lo), _f3(hi), float(edge_scale), float(log_scale_hi),
                           int(n_material_cycle), abi.ptr(out))
    if rc != abi.PT_OK:
        raise PtError(rc, "pt_gen_soup failed")
    return out


def gen_primary_rays(width, height, sensor_width=36.0, focal_length=50.0):
    cam = abi.Camera(width, height, sensor_width, focal_length)
    out = np.zeros(width * height, dtype=abi.RAY)
    rc = lib().pt_gen_primary_rays(C.byref(cam), abi.ptr(out))
    if rc != abi.PT_OK:
        raise PtError(rc, "pt_gen_primary_rays failed")
    return out


class LoadedScene:
    """What Scene::load returns (scene.rs:273-281), as the arrays the C ABI takes."""

    def __init__(self, arrays, camera, camera_transform):
        self.arrays, self.camera, self.camera_transform = arrays, camera, camera_transform


def load_scene_file(path):
    """Scene::load through the library's C++ loader (pt_scene_file_load): TOML + OBJ/MTL + PNG -> arrays.
    Raises ValueError where the reference returns an error or panics."""
    h = C.c_void_p()
    rc = lib().pt_scene_file_load(os.fsencode(str(path)), C.byref(h))
    if rc != abi.PT_OK:
        raise ValueError((lib().pt_last_error(None) or b"").decode())
    try:
        info = abi.SceneFileInfo()
        lib().pt_scene_file_get(h, C.byref(info))
        d = info.desc

        def copy(ptr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer(C.string_at(ptr, n * np.dtype(dtype).itemsize), dtype=dtype).copy()

        tris = copy(d.triangles, d.n_triangles, abi.TRIANGLE)
        nodes = copy(d.nodes, d.n_nodes, abi.BVH_NODE) if d.n_nodes else None
        mats = copy(d.materials, d.n_materials, abi.MATERIAL)
        images = []
        for k in range(d.n_images):
            im = d.images[k]
            tex = copy(im.texels, im.width * im.height * im.channels, np.float32)
            images.append(tex.reshape((im.height, im.width, 3) if im.channels == 3 else (im.height, im.width)))
        sky = abi.Sky.from_buffer_copy(bytes(d.sky))
        cam = {"resolution": (info.camera.width, info.camera.height), "sensor_width": info.camera.sensor_width,
               "focal_length": info.camera.focal_length, "samples": info.samples}
        m = np.array(list(info.camera_transform), dtype=np.float32).reshape(4, 4).T.copy()      # m[row, col]
        return LoadedScene(abi.SceneArrays(tris, nodes, mats, sky, images), cam, m)
    finally:
        lib().pt_scene_file_free(h)


class PinnedArray:
    """A numpy array over pinned host memory from pt_host_alloc (freed with the object)."""

    def __init__(self, shape, dtype):
        self.dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        rc = lib().pt_host_alloc(n, C.byref(self._p))
        if rc != abi.PT_OK:
            raise PtError(rc, (lib().pt_last_error(None) or b"").decode())
        buf = (C.c_char * max(n, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            lib().pt_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- device side ------------------------------------------------------------------------------

class Scene:
    def __init__(self, ctx, handle, arrays):
        self.ctx, self.handle, self.arrays = ctx, handle, arrays

    def info(self):
        i = abi.SceneInfo()
        lib().pt_scene_get_info(self.handle, C.byref(i))
        return i

    def bvh(self):
        """The scene's binary tree as `BVH.nodes` (pt_scene_get_bvh): reference order, also for a device-built tree."""
        n = C.c_uint32(0)
        self.ctx._check(lib().pt_scene_get_bvh(self.ctx.handle, self.handle, None, 0, C.byref(n)))
        nodes = np.zeros(n.value, dtype=abi.BVH_NODE)
        self.ctx._check(lib().pt_scene_get_bvh(self.ctx.handle, self.handle, abi.ptr(nodes) if n.value else None, n.value, C.byref(n)))
        return nodes

    def close(self):
        if self.handle:
            lib().pt_scene_destroy(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One pt_ctx = one GPU."""

    def __init__(self, device=0):
        h = C.c_void_p()
        ids = (C.c_int * 1)(int(device))
        rc = lib().pt_init(ids, 1, C.byref(h))
        if rc != abi.PT_OK:
            raise PtError(rc, (lib().pt_last_error(None) or b"").decode())
        self.handle = h

    def _check(self, rc):
        if rc != abi.PT_OK:
            raise PtError(rc, (lib().pt_last_error(self.handle) or b"").decode())

    def close(self):
        if self.handle:
            lib().pt_shutdown(self.handle)
            self.handle = None

    def scene(self, arrays, lbvh=False):
        """pt_scene_create; lbvh=True builds the BVH on the GPU instead (pt_scene_create_ex, arrays.nodes is ignored)."""
        d = arrays.desc()
        h = C.c_void_p()
        if lbvh:
            d.nodes, d.n_nodes = None, 0
            self._check(lib().pt_scene_create_ex(self.handle, C.byref(d), abi.PT_SCENE_BUILD_LBVH, C.byref(h)))
        else:
            self._check(lib().pt_scene_create(self.handle, C.byref(d), C.byref(h)))
        return Scene(self, h, arrays)

    def set_traversal(self, mode):
        self._check(lib().pt_set_traversal(self.handle, int(mode)))

    def trace(self, scene, rays, start=None, out=None):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY)
        n = rays.size
        if out is None:
            out = np.zeros(n, dtype=abi.HIT)
        assert out.dtype == abi.HIT and out.size == n and out.flags["C_CONTIGUOUS"]
        st = None if start is None else np.ascontiguousarray(start, dtype=np.uint32)
        self._check(lib().pt_trace_closest(self.handle, scene.handle, abi.ptr(rays), abi.ptr(st), n, abi.ptr(out)))
        return out

    def trace_dev(self, scene, d_rays, d_start, n, d_out):
        self._check(lib().pt_trace_closest_dev(self.handle, scene.handle, C.c_void_p(d_rays),
                                               C.c_void_p(d_start) if d_start else None, int(n), C.c_void_p(d_out)))

    def trace_dev_noalloc(self, scene, d_rays, d_start, n, d_out):
        self._check(lib().pt_trace_closest_dev_noalloc(self.handle, scene.handle, C.c_void_p(d_rays),
                                                       C.c_void_p(d_start) if d_start else None, int(n), C.c_void_p(d_out)))

    def trace_dev_submit(self, scene, d_rays, d_start, n, d_out, slot=0):
        """pt_trace_closest_dev without the wait: the batch is queued on `slot` (0 .. abi.PT_TRACE_SLOTS-1)."""
        self._check(lib().pt_trace_closest_dev_submit(self.handle, scene.handle, C.c_void_p(d_rays),
                                                      C.c_void_p(d_start) if d_start else None, int(n), C.c_void_p(d_out), int(slot)))

    def trace_dev_wait(self, slot=0):
        self._check(lib().pt_trace_closest_dev_wait(self.handle, int(slot)))

    def render(self, scene, width, height, spp_begin, spp_count, spp_total, seed, mode=abi.PT_RENDER_FINAL,
               sensor_width=36.0, focal_length=50.0):
        cam = abi.Camera(width, height, sensor_width, focal_length)
        out = np.zeros((height, width, 4), dtype=np.float32)
        self._check(lib().pt_render(self.handle, scene.handle, C.byref(cam), spp_begin, spp_count, spp_total,
                                    int(seed), int(mode), abi.ptr(out)))
        return out

    def render_dev(self, scene, width, height, spp_begin, spp_count, spp_total, seed, mode, d_out,
                   sensor_width=36.0, focal_length=50.0):
        cam = abi.Camera(width, height, sensor_width, focal_length)
        self._check(lib().pt_render_dev(self.handle, scene.handle, C.byref(cam), spp_begin, spp_count, spp_total,
                                        int(seed), int(mode), C.c_void_p(d_out)))

    def tonemap(self, xyzw, avg_zone=5.0, white_zone=10.0, skip_nonfinite=False):
        """renderer.rs:297-309 + shader.wgsl:32-44: returns (linear-sRGB float RGBA, BGRA8 surface bytes,
        (avg_luminance, max_luminance, middle_gray))."""
        xyzw = np.ascontiguousarray(xyzw, dtype=np.float32)
        h, w = xyzw.shape[0], xyzw.shape[1]
        rgba = np.zeros((h, w, 4), dtype=np.float32)
        bgra8 = np.zeros((h, w, 4), dtype=np.uint8)
        stats = (C.c_float * 3)()
        self._check(lib().pt_tonemap(self.handle, abi.ptr(xyzw), w, h, float(avg_zone), float(white_zone),
                                     abi.PT_TONEMAP_SKIP_NONFINITE if skip_nonfinite else 0, abi.ptr(rgba), abi.ptr(bgra8), stats))
        return rgba, bgra8, tuple(stats)

    def stats(self):
        s = abi.Stats()
        lib().pt_get_stats(self.handle, C.byref(s))
        return s

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
