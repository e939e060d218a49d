"""Scene ingestion — the step BEFORE the hot path: the reference's TOML scene file
+ OBJ/MTL meshes -> the arrays pt_scene_create takes.  Host-side mirror of

  scene.rs:64-183    SceneFile / Camera / CameraTransform / Sky / Mesh (serde schema and defaults)
  scene.rs:194-282   Scene::load: mesh.transform * camera_transform, object -> group -> poly order,
                     quads split (a,b,c),(a,c,d), >4-gons rejected, material 0 = Material::default
  scene.rs:16-62     ObjExt::triangle: uv raw, normals through transform_vector, positions through
                     transform_point
  material.rs:39-78,206-253   Material::from_mtl and the image -> texel conversions
  color.rs:6-12,119-133       sRGB decode of texture pixels (flat Kd/Ke are taken as they are)

nalgebra 0.31.3 semantics are restated in binary32 with the same operation order:
Matrix4::from([[f32;4];4]) is column-major, look_at_rh, M*N by column axpy, transform_point divides by
the homogeneous coordinate, transform_vector divides the INPUT by M[3,0..3].v when that is non-zero.

What is not reproduced: the `obj` crate accepts more of the OBJ grammar than this parser (only
v/vt/vn/f/o/g/usemtl/mtllib are read, triangles are emitted in file order, which is the crate's
object -> group -> poly order whenever no object or group name is re-opened), and textures must be
8-bit non-interlaced PNG (the reference decodes any format the `image` crate knows).
Parity unpinned: the reference holds no vectors for its loader; tests/test_scene_io.py checks
hand-derived cases.
"""
import os
import struct
import zlib

import numpy as np

try:
    import tomllib as _toml  # Python >= 3.11
except ImportError:  # pragma: no cover
    import tomli as _toml

from . import abi, core

f32 = np.float32


# ---- nalgebra restatements (binary32, same association order) -----------------------------------

def matrix4_from_columns(cols):
    """Matrix4::from([[f32;4];4]): each inner array is a COLUMN.  Returns m[row, col]."""
    m = np.asarray(cols, dtype=f32)
    if m.shape != (4, 4):
        raise ValueError("a transform must be a 4x4 array")
    return m.T.copy()


def _dot3(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross(a, b):
    return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]),
                     f32(a[0] * b[1]) - f32(a[1] * b[0])], dtype=f32)


def _normalize(v):
    return (v / np.sqrt(_dot3(v, v))).astype(f32)


def look_at_rh(eye, target, up):
    """Matrix4::look_at_rh (scene.rs:165): rows x, y, z of the view basis, translation -(axis . eye)."""
    eye, target, up = (np.asarray(v, dtype=f32) for v in (eye, target, up))
    z = _normalize(eye - target)
    x = _normalize(_cross(up, z))
    y = _cross(z, x)
    m = np.zeros((4, 4), dtype=f32)
    m[0, :3], m[1, :3], m[2, :3] = x, y, z
    m[0, 3], m[1, 3], m[2, 3] = -_dot3(x, eye), -_dot3(y, eye), -_dot3(z, eye)
    m[3, 3] = 1.0
    return m


def mat_mul(a, b):
    """a * b, accumulated over k = 0..3 in order (column axpy)."""
    out = np.zeros((4, 4), dtype=f32)
    for j in range(4):
        col = (a[:, 0] * b[0, j]).astype(f32)
        for k in range(1, 4):
            col = (col + (a[:, k] * b[k, j]).astype(f32)).astype(f32)
        out[:, j] = col
    return out


def _mat3_vec(m, v):
    r = (m[:3, 0] * v[..., 0:1]).astype(f32)
    r = (r + (m[:3, 1] * v[..., 1:2]).astype(f32)).astype(f32)
    return (r + (m[:3, 2] * v[..., 2:3]).astype(f32)).astype(f32)


def _row_dot(m, v):
    return ((m[3, 0] * v[..., 0] + m[3, 1] * v[..., 1]).astype(f32) + m[3, 2] * v[..., 2]).astype(f32)


def transform_points(m, p):
    p = np.asarray(p, dtype=f32).reshape(-1, 3)
    n = (_row_dot(m, p) + m[3, 3]).astype(f32)
    r = (_mat3_vec(m, p) + m[:3, 3]).astype(f32)
    nz = n != 0
    r[nz] = (r[nz] / n[nz, None]).astype(f32)
    return r


def transform_vectors(m, v):
    v = np.asarray(v, dtype=f32).reshape(-1, 3).copy()
    n = _row_dot(m, v)
    nz = n != 0
    v[nz] = (v[nz] / n[nz, None]).astype(f32)
    return _mat3_vec(m, v)


# ---- images (material.rs:39-78, color.rs:6-12,119-133) --------------------------------------------

def to_linear_srgb(u):
    u = np.asarray(u, dtype=f32)
    return np.where(u <= f32(0.04045), u / f32(12.92), np.power((u + f32(0.055)) / f32(1.055), f32(2.4))).astype(f32)


def read_png(path):
    """8-bit, non-interlaced PNG (grey, grey+alpha, RGB, RGBA, palette) -> (uint8 [H, W, C], has_alpha)."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: only PNG textures are supported by this loader")
    pos, idat, palette, trns, hdr = 8, [], None, None, None
    while pos < len(data):
        (length,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif tag == b"tRNS":
            trns = np.frombuffer(body, dtype=np.uint8)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0:
        raise ValueError(f"{path}: only 8-bit non-interlaced PNG is supported")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(w * ch, dtype=np.int32)
            for i in range(w * ch):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, ch)
    if ctype == 3:
        rgb = palette[img[..., 0]]
        if trns is not None:
            alpha = np.full(256, 255, dtype=np.uint8)
            alpha[:trns.size] = trns
            return np.concatenate([rgb, alpha[img[..., 0]][..., None]], axis=-1), True
        return rgb, False
    if ctype == 0:
        return np.repeat(img, 3, axis=-1), False
    if ctype == 4:
        return np.concatenate([np.repeat(img[..., :1], 3, axis=-1), img[..., 1:]], axis=-1), True
    return img, ctype == 6


def image_color(path):
    """Image<Color<SRGB>> (material.rs:39-52): linear-sRGB texels [H, W, 3]."""
    px, _ = read_png(path)
    return to_linear_srgb(px[..., :3].astype(f32) * f32(1.0 / 255.0))


def image_scalar(path):
    """Image<f32> (material.rs:54-78): alpha/255 when the file has alpha, else luminance of the decoded colour."""
    px, has_alpha = read_png(path)
    if has_alpha:
        return (px[..., 3].astype(f32) / f32(255.0)).astype(f32)
    lin = to_linear_srgb(px[..., :3].astype(f32) * f32(1.0 / 255.0))
    y = (f32(0.21263901) * lin[..., 0] + f32(0.71516868) * lin[..., 1]).astype(f32) + f32(0.07219232) * lin[..., 2]
    return y.astype(f32)


# ---- OBJ / MTL ----------------------------------------------------------------------------------------

def parse_mtl(path):
    mats, cur = {}, None
    for line in open(path, errors="replace"):
        t = line.split("#", 1)[0].split()
        if not t:
            continue
        key = t[0]
        if key == "newmtl":
            cur = {"name": " ".join(t[1:])}
            mats[cur["name"]] = cur
        elif cur is None:
            continue
        elif key in ("Kd", "Ke"):
            cur[key.lower()] = [float(x) for x in t[1:4]]
        elif key == "d":
            cur["d"] = float(t[1])
        elif key in ("map_Kd", "map_Ka", "map_d"):
            cur[key.lower()] = t[-1]
    return mats


def parse_obj(path):
    """-> positions, texcoords, normals, faces [(verts [(v, vt, vn)], material name or None)], mtllibs."""
    pos, tex, nrm, faces, libs, mat = [], [], [], [], [], None

    def idx(s, n):
        if s == "":
            return None
        i = int(s)
        return i - 1 if i > 0 else n + i

    for line in open(path, errors="replace"):
        t = line.split("#", 1)[0].split()
        if not t:
            continue
        if t[0] == "v":
            pos.append([float(x) for x in t[1:4]])
        elif t[0] == "vt":
            tex.append([float(t[1]), float(t[2]) if len(t) > 2 else 0.0])
        elif t[0] == "vn":
            nrm.append([float(x) for x in t[1:4]])
        elif t[0] == "usemtl":
            mat = " ".join(t[1:])
        elif t[0] == "mtllib":
            libs.extend(t[1:])
        elif t[0] == "f":
            verts = []
            for v in t[1:]:
                p = (v.split("/") + ["", ""])[:3]
                verts.append((idx(p[0], len(pos)), idx(p[1], len(tex)), idx(p[2], len(nrm))))
            faces.append((verts, mat))
    return (np.asarray(pos, dtype=f32).reshape(-1, 3), np.asarray(tex, dtype=f32).reshape(-1, 2),
            np.asarray(nrm, dtype=f32).reshape(-1, 3), faces, libs)


class LoadedScene:
    """What Scene::load returns, as the arrays the C ABI takes."""

    def __init__(self, arrays, camera, camera_transform):
        self.arrays, self.camera, self.camera_transform = arrays, camera, camera_transform


def _material_from_mtl(mtl, mtl_dir, images):
    """Material::from_mtl (material.rs:206-253)."""
    def add_image(texels):
        images.append(np.ascontiguousarray(texels, dtype=f32))
        return len(images) - 1

    m = abi.default_material()
    if "map_kd" in mtl:
        m["diffuse"]["kind"], m["diffuse"]["image"] = abi.PT_TEX_NEAREST, add_image(image_color(os.path.join(mtl_dir, mtl["map_kd"])))
    elif "kd" in mtl:
        m["diffuse"]["value"] = mtl["kd"]                       # Color::srgb(kd): taken as linear, not decoded
    if "map_ka" in mtl:
        m["emit_kind"] = abi.PT_EMIT_ABSORBED_D65
        m["absorption"]["kind"], m["absorption"]["image"] = abi.PT_TEX_NEAREST, add_image(image_color(os.path.join(mtl_dir, mtl["map_ka"])))
    elif "ke" in mtl:
        m["emit_kind"] = abi.PT_EMIT_ABSORBED_D65
        m["absorption"]["value"] = mtl["ke"]
    m["base_dissolve"] = mtl.get("d", 1.0)
    if "map_d" in mtl:
        m["dissolve"]["kind"], m["dissolve"]["image"] = abi.PT_TEX_NEAREST, add_image(image_scalar(os.path.join(mtl_dir, mtl["map_d"])))
    return m


def load_scene(path):
    """Scene::load (scene.rs:194-282).  Returns a LoadedScene; raises ValueError where the reference
    returns an error or panics."""
    with open(path, "rb") as f:
        doc = _toml.load(f)
    cam = {"resolution": (1024, 768), "sensor_width": 36.0, "focal_length": 50.0, "samples": 128}     # scene.rs:142-152
    cam_doc = doc.get("camera", {})
    for k in cam:
        if k in cam_doc:
            cam[k] = tuple(cam_doc[k]) if k == "resolution" else cam_doc[k]
    tr = cam_doc.get("transform")
    if tr is None:
        camera_transform = np.eye(4, dtype=f32)
    elif isinstance(tr, dict):                                   # CameraTransform::LookAt (untagged, tried first)
        if not all(k in tr for k in ("eye", "target", "up")):
            raise ValueError("camera.transform: expected {eye, target, up} or a 4x4 array")
        camera_transform = look_at_rh(tr["eye"], tr["target"], tr["up"])
    else:
        camera_transform = matrix4_from_columns(tr)

    triangles, materials, images = [], [abi.default_material()], []
    base = os.path.dirname(os.path.abspath(path))
    for mesh in doc.get("meshes", []):
        if "path" not in mesh:
            raise ValueError("meshes: `path` is required")
        mesh_tf = matrix4_from_columns(mesh["transform"]) if "transform" in mesh else np.eye(4, dtype=f32)
        transform = mat_mul(mesh_tf, camera_transform)          # scene.rs:211
        mesh_path = os.path.join(base, mesh["path"])              # path.with_file_name(mesh.path)
        pos, tex, nrm, faces, libs = parse_obj(mesh_path)
        mtls = {}
        for lib in libs:
            lib_path = os.path.join(os.path.dirname(mesh_path), lib)
            for name, m in parse_mtl(lib_path).items():
                mtls.setdefault(name, (m, os.path.dirname(lib_path)))
        tpos = transform_points(transform, pos) if pos.size else pos
        tnrm = transform_vectors(transform, nrm) if nrm.size else nrm
        material_map = {}
        for verts, mat_name in faces:
            if mat_name is None or mat_name not in mtls:          # None, or ObjMaterial::Ref ("missing material")
                material = 0
            else:
                if mat_name not in material_map:
                    material_map[mat_name] = len(materials)
                    materials.append(_material_from_mtl(mtls[mat_name][0], mtls[mat_name][1], images))
                material = material_map[mat_name]
            if len(verts) < 3:
                continue                                          # "Ignoring ..." (scene.rs:250-252)
            if len(verts) > 4:
                raise ValueError(f"unhandled {len(verts)}-gon")   # scene.rs:262-264
            corners = [(0, 1, 2)] if len(verts) == 3 else [(0, 1, 2), (0, 2, 3)]        # scene.rs:253-261
            for tri in corners:
                a, b, c = (verts[i] for i in tri)
                t = np.zeros((), dtype=abi.TRIANGLE)
                t["a"], t["b"], t["c"] = tpos[a[0]], tpos[b[0]], tpos[c[0]]
                has_uv = [v[1] is not None for v in (a, b, c)]
                if any(has_uv) and not all(has_uv):
                    raise ValueError("texture coords defined on some vertices but not all")     # scene.rs:43
                has_n = [v[2] is not None for v in (a, b, c)]
                if any(has_n) and not all(has_n):
                    raise ValueError("normals defined on some vertices but not all")            # scene.rs:53
                flags = 0
                if all(has_uv):
                    t["uv"] = [tex[a[1]], tex[b[1]], tex[c[1]]]
                    flags |= abi.PT_TRI_HAS_UV
                if all(has_n):
                    t["vertex_normals"] = [tnrm[a[2]], tnrm[b[2]], tnrm[c[2]]]
                    flags |= abi.PT_TRI_HAS_NORMALS
                t["flags"], t["material_index"] = flags, material
                triangles.append(t)

    sky_doc = doc.get("sky")
    if sky_doc is None:
        sky = core.sky_d65(0.0)                                   # Sky::default, scene.rs:88-92
    elif "power" in sky_doc:                                      # untagged: D65 is tried first
        sky = core.sky_d65(float(sky_doc["power"]))
    else:
        try:
            args = [sky_doc[k] for k in ("atmospheric_turbidity", "ground_albedo", "zenith_direction", "sun_direction")]
        except KeyError as e:
            raise ValueError(f"sky: missing {e}") from None
        try:
            sky = core.sky_hosek_wilkie(args[0], args[1], args[2], args[3], camera_transform.T.copy())   # column-major
        except core.PtError as e:
            raise ValueError(str(e)) from None

    tris = np.array(triangles, dtype=abi.TRIANGLE) if triangles else np.zeros(0, dtype=abi.TRIANGLE)
    nodes = core.bvh_build(tris) if tris.size else None
    arrays = abi.SceneArrays(tris, nodes, np.array(materials, dtype=abi.MATERIAL), sky, images)
    return LoadedScene(arrays, cam, camera_transform)
