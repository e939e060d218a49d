"""Scene ingestion (pathtracer_amd/scene_io.py) against hand-derived cases of the
reference's loader (scene.rs:16-62,64-183,194-282; material.rs:39-78,206-253).  The reference holds
no vectors for its loader, so these are checks of the restatement, not of the reference binary."""
import struct
import zlib

import numpy as np
import pytest

from pathtracer_amd import abi, scene_io, workloads

OBJ = """\
mtllib mats.mtl
v -1 -1 -3
v  1 -1 -3
v  1  1 -3
v -1  1 -3
v  0  0 -2
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
o wall
usemtl red
f 1/1/1 2/2/1 3/3/1 4/4/1
usemtl lamp
f 1 2 5
usemtl nosuch
f -3//1 -2//1 -1//1
g loose
f 1 2
usemtl red
f 2/2 3/3 5/1
"""
MTL = """\
newmtl red
Kd 0.8 0.1 0.1
d 0.5
newmtl lamp
Ke 1.0 0.9 0.8
map_Kd tex.png
map_d mask.png
"""


def png_bytes(px):
    px = np.asarray(px, dtype=np.uint8)
    h, w, c = px.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.fixture
def scene_dir(tmp_path):
    (tmp_path / "m.obj").write_text(OBJ)
    (tmp_path / "mats.mtl").write_text(MTL)
    tex = np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [128, 128, 128]]], dtype=np.uint8)
    (tmp_path / "tex.png").write_bytes(png_bytes(tex))
    mask = np.array([[[10, 20, 30, 255], [10, 20, 30, 0]]], dtype=np.uint8)
    (tmp_path / "mask.png").write_bytes(png_bytes(mask))
    return tmp_path


def test_defaults_of_an_empty_scene_file(tmp_path):
    (tmp_path / "s.toml").write_text("")
    s = scene_io.load_scene(tmp_path / "s.toml")
    assert s.camera == {"resolution": (1024, 768), "sensor_width": 36.0, "focal_length": 50.0, "samples": 128}   # scene.rs:142-152
    assert s.arrays.triangles.size == 0 and s.arrays.nodes is None
    assert s.arrays.sky.kind == abi.PT_SKY_D65 and s.arrays.sky.power == 0.0                                     # scene.rs:88-92
    assert s.arrays.materials.size == 1 and s.arrays.materials[0].tobytes() == abi.default_material().tobytes()  # scene.rs:204


def test_obj_mtl_ingestion(scene_dir):
    (scene_dir / "s.toml").write_text('[camera]\nresolution = [64, 48]\nsamples = 4\n[sky]\npower = 1.0\n[[meshes]]\npath = "m.obj"\n')
    s = scene_io.load_scene(scene_dir / "s.toml")
    t = s.arrays.triangles
    # quad -> (a,b,c),(a,c,d); then the lamp triangle; the `nosuch` triangle (negative indices); the
    # 2-vertex face is ignored; last triangle has uv but no normals
    assert t.size == 5
    assert np.array_equal(t["a"][0], [-1, -1, -3]) and np.array_equal(t["b"][0], [1, -1, -3]) and np.array_equal(t["c"][0], [1, 1, -3])
    assert np.array_equal(t["a"][1], [-1, -1, -3]) and np.array_equal(t["b"][1], [1, 1, -3]) and np.array_equal(t["c"][1], [-1, 1, -3])
    assert np.array_equal(t["uv"][1], [[0, 0], [1, 1], [0, 1]])
    assert t["flags"].tolist() == [3, 3, 0, 2, 1]
    assert np.array_equal(t["a"][3], [1, 1, -3]) and np.array_equal(t["c"][3], [0, 0, -2])    # -3, -2, -1 of five vertices
    # materials: 0 default, 1 red (first use), 2 lamp; unknown name -> 0 (scene.rs:241-244)
    assert t["material_index"].tolist() == [1, 1, 2, 0, 1]
    m = s.arrays.materials
    assert m.size == 3
    np.testing.assert_allclose(m[1]["diffuse"]["value"], [0.8, 0.1, 0.1])      # Kd is NOT sRGB-decoded (material.rs:217)
    assert m[1]["base_dissolve"] == np.float32(0.5) and m[1]["emit_kind"] == abi.PT_EMIT_NULL
    assert m[2]["emit_kind"] == abi.PT_EMIT_ABSORBED_D65 and m[2]["diffuse"]["kind"] == abi.PT_TEX_NEAREST
    assert m[2]["dissolve"]["kind"] == abi.PT_TEX_NEAREST and m[2]["base_dissolve"] == 1.0
    np.testing.assert_allclose(m[2]["absorption"]["value"], [1.0, 0.9, 0.8])
    # images: colour texture decoded to linear sRGB, row-major from the top; dissolve = alpha / 255
    col, mask = s.arrays.images[m[2]["diffuse"]["image"]], s.arrays.images[m[2]["dissolve"]["image"]]
    assert col.shape == (2, 2, 3) and mask.shape == (1, 2)
    assert col[0, 0].tolist() == [1.0, 0.0, 0.0] and col[1, 0].tolist() == [0.0, 0.0, 1.0]
    want = ((128 / 255 + 0.055) / 1.055) ** 2.4
    np.testing.assert_allclose(col[1, 1], want, rtol=1e-6)
    assert mask.tolist() == [[1.0, 0.0]]
    assert s.arrays.nodes.size == 2 * t.size - 1 and s.arrays.sky.power == 1.0


def test_transform_order_and_column_major(scene_dir):
    """transform = mesh.transform * camera_transform (scene.rs:211); [[f32;4];4] arrays are COLUMNS."""
    cam_t = "[[1,0,0,0],[0,1,0,0],[0,0,1,0],[0,0,-5,1]]"         # translate z by -5 (4th COLUMN)
    mesh_t = "[[2,0,0,0],[0,2,0,0],[0,0,2,0],[0,0,0,1]]"         # scale by 2
    (scene_dir / "s.toml").write_text(f'[camera]\ntransform = {cam_t}\n[[meshes]]\npath = "m.obj"\ntransform = {mesh_t}\n')
    s = scene_io.load_scene(scene_dir / "s.toml")
    # camera transform is applied FIRST, then the mesh transform: (-1,-1,-3) -> (-1,-1,-8) -> (-2,-2,-16)
    assert np.array_equal(s.arrays.triangles["a"][0], [-2, -2, -16])
    # normals go through transform_vector: scaled, not translated, not re-normalised
    assert np.array_equal(s.arrays.triangles["vertex_normals"][0][0], [0, 0, 2])


def test_look_at_and_sky_in_camera_space(scene_dir):
    (scene_dir / "s.toml").write_text(
        '[camera]\ntransform = { eye = [0, 0, 5], target = [0, 0, 0], up = [0, 1, 0] }\n'
        '[sky]\natmospheric_turbidity = 3.0\nground_albedo = 0.3\nzenith_direction = [0, 1, 0]\nsun_direction = [0.3, 0.8, 0.5]\n'
        '[[meshes]]\npath = "m.obj"\n')
    s = scene_io.load_scene(scene_dir / "s.toml")
    # looking down -Z from z = 5: a pure translation by -5 in z
    np.testing.assert_allclose(s.camera_transform, [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -5], [0, 0, 0, 1]], atol=1e-7)
    assert np.array_equal(s.arrays.triangles["a"][0], [-1, -1, -8])
    assert s.arrays.sky.kind == abi.PT_SKY_HOSEK_WILKIE
    np.testing.assert_allclose(list(s.arrays.sky.zenith), [0, 1, 0], atol=1e-7)
    # a camera that looks along +X: the world's up stays up, the sun direction rotates with the view
    (scene_dir / "t.toml").write_text(
        '[camera]\ntransform = { eye = [0, 0, 0], target = [1, 0, 0], up = [0, 1, 0] }\n'
        '[sky]\natmospheric_turbidity = 3.0\nground_albedo = 0.3\nzenith_direction = [0, 1, 0]\nsun_direction = [1, 1, 0]\n')
    s2 = scene_io.load_scene(scene_dir / "t.toml")
    np.testing.assert_allclose(list(s2.arrays.sky.sun), [0, 2 ** -0.5, -(2 ** -0.5)], atol=1e-6)   # straight ahead and up


def test_loader_errors(scene_dir):
    (scene_dir / "five.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nf 1 2 3 4 5\n")
    (scene_dir / "a.toml").write_text('[[meshes]]\npath = "five.obj"\n')
    with pytest.raises(ValueError, match="5-gon"):
        scene_io.load_scene(scene_dir / "a.toml")
    (scene_dir / "part.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nvt 0 0\nf 1/1 2 3\n")
    (scene_dir / "b.toml").write_text('[[meshes]]\npath = "part.obj"\n')
    with pytest.raises(ValueError, match="texture coords"):
        scene_io.load_scene(scene_dir / "b.toml")
    (scene_dir / "c.toml").write_text('[sky]\natmospheric_turbidity = 3.0\nground_albedo = 0.3\nzenith_direction = [0,1,0]\nsun_direction = [0,-1,0]\n')
    with pytest.raises(ValueError):           # assert!(solar_elevation > 0.0), scene.rs:121
        scene_io.load_scene(scene_dir / "c.toml")
    (scene_dir / "d.toml").write_text('[sky]\n')
    with pytest.raises(ValueError):           # neither untagged variant matches
        scene_io.load_scene(scene_dir / "d.toml")


def test_png_reader_filters(tmp_path):
    """All five PNG filter types, written by hand."""
    rng = np.random.default_rng(1)
    px = rng.integers(0, 256, size=(5, 6, 3), dtype=np.uint8)
    bpp, stride = 3, 18
    raw, prev = b"", np.zeros(stride, dtype=np.int32)
    for y in range(5):
        line = px[y].reshape(-1).astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), line[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if y == 0:
            enc = line
        elif y == 1:
            enc = line - left
        elif y == 2:
            enc = line - prev
        elif y == 3:
            enc = line - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = line - pred
        raw += bytes([y]) + (enc & 255).astype(np.uint8).tobytes()
        prev = line

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    (tmp_path / "f.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 6, 5, 8, 2, 0, 0, 0))
                                     + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    got, has_alpha = scene_io.read_png(tmp_path / "f.png")
    assert not has_alpha and np.array_equal(got, px)


# ---- the library's C++ loader (pt_scene_file_load) against the Python restatement of the same rules ----

def same_loaded_scene(a, b):
    assert a.camera == b.camera
    assert a.camera_transform.tobytes() == b.camera_transform.tobytes()
    assert a.arrays.triangles.tobytes() == b.arrays.triangles.tobytes()
    assert (a.arrays.nodes is None) == (b.arrays.nodes is None)
    if a.arrays.nodes is not None:
        assert a.arrays.nodes.tobytes() == b.arrays.nodes.tobytes()
    assert a.arrays.materials.tobytes() == b.arrays.materials.tobytes()
    assert bytes(a.arrays.sky) == bytes(b.arrays.sky)
    assert len(a.arrays.images) == len(b.arrays.images)
    for x, y in zip(a.arrays.images, b.arrays.images):
        assert x.shape == y.shape
        np.testing.assert_allclose(x, y, rtol=3e-7, atol=0)      # powf of two libms: 1-2 ulp


SCENES = {
    "empty": "",
    "basic": '[camera]\nresolution = [64, 48]\nsamples = 4\n[sky]\npower = 1.0\n[[meshes]]\npath = "m.obj"\n',
    "matrices": '[camera]\ntransform = [[1,0,0,0],[0,1,0,0],[0,0,1,0],[0,0,-5,1]]\n[[meshes]]\npath = "m.obj"\n'
                'transform = [[2,0,0,0],[0,2,0,0],[0,0,2,0],[0,0,0,1]]\n',
    "look_at_hw": '# a comment\n[camera]\nsensor_width = 24.0 # trailing\nfocal_length = 35\n'
                  'transform = { eye = [0.5, 1.25, 5], target = [0, 0.1, 0], up = [0, 1, 0] }\n'
                  '[sky]\natmospheric_turbidity = 3.0\nground_albedo = 0.3\nzenith_direction = [0, 1, 0]\n'
                  'sun_direction = [\n  0.3,\n  0.8, # inside an array\n  0.5,\n]\n'
                  '[[meshes]]\npath = "m.obj"\n[[meshes]]\npath = \'m.obj\'\ntransform = [[1,0,0,0],[0,1,0,0],[0,0,1,0],[0.25,0,-1,1]]\n',
    "dotted_table": '[camera.transform]\neye = [0, 0, 5]\ntarget = [0, 0, 0]\nup = [0, 1, 0]\n[camera]\nsamples = 1_000\n',
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_cpp_loader_agrees_with_the_python_restatement(scene_dir, name):
    from pathtracer_amd import core
    (scene_dir / "s.toml").write_text(SCENES[name])
    same_loaded_scene(core.load_scene_file(scene_dir / "s.toml"), scene_io.load_scene(scene_dir / "s.toml"))


def test_cpp_loader_errors(scene_dir):
    from pathtracer_amd import core
    (scene_dir / "five.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nf 1 2 3 4 5\n")
    (scene_dir / "part.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nvt 0 0\nf 1/1 2 3\n")
    cases = {
        '[[meshes]]\npath = "five.obj"\n': "5-gon",
        '[[meshes]]\npath = "part.obj"\n': "texture coords",
        '[[meshes]]\npath = "nosuch.obj"\n': "cannot open",
        '[[meshes]]\n': "path",
        '[sky]\natmospheric_turbidity = 3.0\nground_albedo = 0.3\nzenith_direction = [0,1,0]\nsun_direction = [0,-1,0]\n': "elevation",
        '[sky]\n': "sky",
        '[camera]\nresolution = [64]\n': "resolution",
        '[camera]\nsamples = "many"\n': "samples",
        '[camera\n': "TOML line 1",
        'a = 1\na = 2\n': "duplicate",
    }
    for text, needle in cases.items():
        (scene_dir / "e.toml").write_text(text)
        with pytest.raises(ValueError, match=needle):
            core.load_scene_file(scene_dir / "e.toml")
    with pytest.raises(ValueError, match="cannot open"):
        core.load_scene_file(scene_dir / "absent.toml")


def test_cpp_png_reader(tmp_path):
    """The library's own inflate + PNG defilter: dynamic-Huffman, fixed-Huffman and stored streams, palette + tRNS."""
    from pathtracer_amd import core
    rng = np.random.default_rng(5)
    smooth = (np.add.outer(np.arange(40), np.arange(56))[..., None] * np.array([1, 2, 3]) % 256).astype(np.uint8)
    noisy = rng.integers(0, 256, size=(9, 7, 4), dtype=np.uint8)
    grey_alpha = rng.integers(0, 256, size=(5, 5, 2), dtype=np.uint8)

    def write(name, px, level):
        h, w, c = px.shape
        ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
        raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))

        def chunk(tag, data):
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
        comp = zlib.compress(raw, level)
        (tmp_path / name).write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
                                      + chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))

    # palette image with a tRNS chunk (colour type 3)
    idx = rng.integers(0, 4, size=(6, 5, 1), dtype=np.uint8)
    raw = b"".join(b"\x00" + idx[y].tobytes() for y in range(6))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    (tmp_path / "pal.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 5, 6, 8, 3, 0, 0, 0))
                                       + chunk(b"PLTE", bytes([255, 0, 0, 0, 255, 0, 0, 0, 255, 90, 90, 90]))
                                       + chunk(b"tRNS", bytes([0, 128])) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    write("smooth.png", smooth, 9)        # dynamic Huffman with long matches
    write("noisy.png", noisy, 0)          # stored blocks
    write("ga.png", grey_alpha, 1)
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 -1\nv 1 0 -1\nv 0 1 -1\nusemtl a\nf 1 2 3\nusemtl b\nf 1 2 3\nusemtl c\nf 1 2 3\nusemtl d\nf 1 2 3\n")
    (tmp_path / "m.mtl").write_text("newmtl a\nmap_Kd smooth.png\nmap_d smooth.png\nnewmtl b\nmap_Ka noisy.png\nmap_d noisy.png\nnewmtl c\nmap_Kd ga.png\nmap_d ga.png\nnewmtl d\nmap_Kd pal.png\nmap_d pal.png\n")
    (tmp_path / "s.toml").write_text('[[meshes]]\npath = "m.obj"\n')
    same_loaded_scene(core.load_scene_file(tmp_path / "s.toml"), scene_io.load_scene(tmp_path / "s.toml"))


@pytest.mark.gpu
def test_loaded_scene_renders_like_the_oracle(oracle, gpu_ctx, scene_dir):
    (scene_dir / "s.toml").write_text('[camera]\nresolution = [64, 48]\nsamples = 4\n[sky]\npower = 1.0\n[[meshes]]\npath = "m.obj"\n')
    s = scene_io.load_scene(scene_dir / "s.toml")
    w, h = s.camera["resolution"]
    got = gpu_ctx.render(gpu_ctx.scene(s.arrays), w, h, 0, s.camera["samples"], s.camera["samples"], seed=1,
                         sensor_width=s.camera["sensor_width"], focal_length=s.camera["focal_length"])
    want = oracle.render(oracle.scene(s.arrays), w, h, 0, 4, 4, seed=1)
    assert got.tobytes() == want.tobytes() and got[..., 1].max() > 0
