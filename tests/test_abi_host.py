"""The C-ABI library without a GPU: it loads, exports every symbol include/ptcore.h declares, its
plain-data structs have the sizes the numpy/ctypes mirrors assume, and the host-side mirrors
(pt_bvh_build, pt_sky_hosek_wilkie, generators) agree with the CPU oracle bit for bit.
No compute entry point is called here.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from pathtracer_amd import abi, core, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptcore.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 19 and "pt_trace_closest" in names and "pt_render" in names
    L = C.CDLL(os.path.join(ROOT, "pathtracer_amd", "libptcore.so"))
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in ptcore.h but not exported: {missing}"
    assert set(core.EXPORTS) == set(names), "pathtracer_amd.core binds a different set than the header declares"
    assert core.lib().pt_abi_version() == 4


def test_struct_layouts_match_the_header(tmp_path):
    """Compile a C (not C++) program against the header: the header is plain C and the sizes/offsets
    are what abi.py mirrors."""
    src = tmp_path / "sizes.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "ptcore.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(pt_aabb), sizeof(pt_bvh_node), sizeof(pt_triangle),
         sizeof(pt_ray), sizeof(pt_hit), sizeof(pt_texture), sizeof(pt_material), sizeof(pt_sky), sizeof(pt_camera),
         sizeof(pt_scene_desc), sizeof(pt_stats), sizeof(pt_scene_info));
  printf("%zu %zu %zu %zu\n", offsetof(pt_triangle, material_index), offsetof(pt_bvh_node, child_r_aabb),
         offsetof(pt_sky, zenith), offsetof(pt_scene_desc, sky));
  return 0;
}''')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = list(map(int, out))
    want = [abi.AABB.itemsize, abi.BVH_NODE.itemsize, abi.TRIANGLE.itemsize, abi.RAY.itemsize, abi.HIT.itemsize,
            abi.TEXTURE.itemsize, abi.MATERIAL.itemsize, C.sizeof(abi.Sky), C.sizeof(abi.Camera),
            C.sizeof(abi.SceneDesc), C.sizeof(abi.Stats), C.sizeof(abi.SceneInfo)]
    assert sizes[:12] == want
    assert sizes[12] == abi.TRIANGLE.fields["material_index"][1]
    assert sizes[13] == abi.BVH_NODE.fields["child_r_aabb"][1]
    assert sizes[14] == abi.Sky.zenith.offset and sizes[15] == abi.SceneDesc.sky.offset


def test_no_gpu_means_a_loud_error_not_a_fallback():
    """Without a HIP device pt_init must fail with an error code and text; nothing renders on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(core.PtError) as e:
        core.Context(0)
    assert e.value.code in (abi.PT_ERR_HIP, abi.PT_ERR_INVALID)
    assert "pt_init" in str(e.value)


# ---- host mirrors vs the oracle ------------------------------------------------------------------

@pytest.mark.parametrize("n", [1, 2, 3, 7, 100, 5000])
def test_bvh_build_is_bit_identical_to_the_oracle(oracle, n):
    """pt_bvh_build (product, pathtracer_amd/csrc/host_bvh.cpp) == the oracle's restatement of
    BVHNode::build (bvh.rs:134-248): same preorder node vector, same boxes, bit for bit."""
    tris = workloads.soup_triangles(n)
    got = core.bvh_build(tris)
    arr = abi.SceneArrays(tris, None, workloads.soup_materials(), core.sky_d65(0.0))
    want = oracle.scene(arr).nodes()
    assert got.size == 2 * n - 1
    assert got.tobytes() == want.tobytes()


def test_bvh_build_long_thin_triangles_and_coincident_centroids(oracle):
    # config-5 style: log-uniform edge scale -> heavily overlapping boxes
    tris = core.gen_soup(3, 3000, (-4, -4, -12), (4, 4, -4), 0.002, 0.2, 7)
    arr = abi.SceneArrays(tris, None, workloads.soup_materials(), core.sky_d65(0.0))
    assert core.bvh_build(tris).tobytes() == oracle.scene(arr).nodes().tobytes()
    # identical triangles: centroid extent < EPSILON -> the index list is halved (bvh.rs:171-183)
    same = np.repeat(workloads.soup_triangles(1), 9)
    nodes = core.bvh_build(same)
    arr = abi.SceneArrays(same, None, workloads.soup_materials(), core.sky_d65(0.0))
    assert nodes.tobytes() == oracle.scene(arr).nodes().tobytes()
    assert nodes.size == 17 and sorted(nodes["shape_index"][nodes["is_leaf"] == 1].tolist()) == list(range(9))


def test_bvh_invariants():
    tris = workloads.soup_triangles(2000)
    nodes = core.bvh_build(tris)
    leaf = nodes["is_leaf"] == 1
    # one triangle per leaf, every triangle once, never permuted (bvh.rs:134-138,155-158)
    assert sorted(nodes["shape_index"][leaf].tolist()) == list(range(2000))
    idx = np.arange(nodes.size)
    inner = ~leaf
    # preorder: left child directly follows its parent, right child comes later (bvh.rs:163-164,236-237)
    assert (nodes["child_l_index"][inner] == idx[inner] + 1).all()
    assert (nodes["child_r_index"][inner] > nodes["child_l_index"][inner]).all()
    # a parent's child box contains the triangle boxes below it (spot-check the root)
    lo = np.minimum(np.minimum(tris["a"], tris["b"]), tris["c"]).min(axis=0)
    hi = np.maximum(np.maximum(tris["a"], tris["b"]), tris["c"]).max(axis=0)
    root = nodes[0]
    assert np.array_equal(np.minimum(root["child_l_aabb"]["min"], root["child_r_aabb"]["min"]), lo)
    assert np.array_equal(np.maximum(root["child_l_aabb"]["max"], root["child_r_aabb"]["max"]), hi)


def test_bvh_build_degenerate_input_is_an_error_code():
    """NaN coordinates make BVH::build hit assert!(!child_aabb.is_empty()) / an out-of-range bucket
    in the reference (bvh.rs:202,242-243): the ABI returns a code instead of panicking."""
    tris = workloads.soup_triangles(16)
    tris["a"][3] = np.nan
    out = np.zeros(31, dtype=abi.BVH_NODE)
    cnt = C.c_uint32(0)
    rc = core.lib().pt_bvh_build(abi.ptr(tris), 16, abi.ptr(out), C.byref(cnt))
    assert rc in (abi.PT_ERR_BVH_DEGENERATE, abi.PT_ERR_INVALID)


def _sky_bytes(s):
    return bytes(C.string_at(C.addressof(s), C.sizeof(s)))


@pytest.mark.parametrize("turb,alb,sun", [(3.0, 0.3, (0.3, 0.8, 0.5)), (1.0, 0.0, (0.0, 0.2, -1.0)),
                                          (10.0, 1.0, (1.0, 1.0, 1.0)), (4.7, 0.55, (-0.4, 0.1, 0.2))])
def test_sky_init_matches_the_oracle(oracle, turb, alb, sun):
    """pt_sky_hosek_wilkie (product) == InitialisedSky::from_sky + HosekWilkieSkyModel::new restated
    in the oracle (scene.rs:109-127, hosek_wilkie.rs:46-72,144-304)."""
    got = core.sky_hosek_wilkie(turb, alb, (0.0, 1.0, 0.0), sun)
    rc, want = oracle.sky_hosek_wilkie(turb, alb, (0.0, 1.0, 0.0), sun)
    assert rc == abi.PT_OK and _sky_bytes(got) == _sky_bytes(want)
    assert got.kind == abi.PT_SKY_HOSEK_WILKIE and abs(got.solar_radius - np.radians(0.51) / 2) < 1e-9
    assert all(r > 0 for r in got.radiances)
    # with a camera transform: look_at style rotation about Y by 90 degrees (column-major)
    m = np.array([[0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    got = core.sky_hosek_wilkie(turb, alb, (0.0, 1.0, 0.0), sun, m)
    rc, want = oracle.sky_hosek_wilkie(turb, alb, (0.0, 1.0, 0.0), sun, m)
    assert rc == abi.PT_OK and _sky_bytes(got) == _sky_bytes(want)


def test_sky_init_rejects_what_the_reference_panics_on():
    with pytest.raises(core.PtError):       # assert!(solar_elevation > 0.0) scene.rs:121
        core.sky_hosek_wilkie(3.0, 0.3, (0, 1, 0), (0.3, -0.1, 0.5))
    with pytest.raises(core.PtError):       # turbidity range hosek_wilkie.rs:348-353
        core.sky_hosek_wilkie(11.0, 0.3, (0, 1, 0), (0.3, 0.8, 0.5))


def test_default_material_mirror():
    m = np.zeros((), dtype=abi.MATERIAL)
    core.lib().pt_material_default(m.ctypes.data_as(C.c_void_p))
    assert m.tobytes() == abi.default_material().tobytes()      # material.rs:300-309


def test_soup_generator_is_deterministic_and_in_range():
    a = workloads.soup_triangles(1000)
    b = workloads.soup_triangles(1000)
    assert a.tobytes() == b.tobytes()
    assert a["a"][:, :2].min() >= -1 and a["a"][:, :2].max() <= 1 and a["a"][:, 2].min() >= -4 and a["a"][:, 2].max() <= -2
    s = 2.0 / 10.0
    assert np.abs(a["b"] - a["a"]).max() <= s * 1.0001 and np.abs(a["c"] - a["a"]).max() <= s * 1.0001
    assert sorted(set(a["material_index"].tolist())) == list(range(1, 9))
    # a prefix of a longer soup is not required to match; a different seed must differ
    assert workloads.soup_triangles(1000, seed=1).tobytes() != a.tobytes()


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under pathtracer_amd/ may import, load or link it."""
    pkg = os.path.join(ROOT, "pathtracer_amd")
    hits = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".S")) or f == "Makefile":
                text = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"libptoracle|pt_oracle|tests\.oracle|\boracle/", text):
                    hits.append(os.path.join(d, f))
    assert not hits, hits
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libptcore.so")]).decode()
    assert "ptoracle" not in out


def test_integration_doc_declares_every_entry_point():
    """INTEGRATION.md shows the reference-side binding: every function of include/ptcore.h appears in it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "ptcore.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    assert [n for n in names if n not in doc] == []


def test_bvh_build_randomised_against_the_recursive_restatement(oracle):
    """Adversarial inputs for the binned-SAH decisions: centroids in a plane, all at one point (the `extent < EPSILON`
    halving rule, bvh.rs:171-183), on a coarse grid (many exact ties between buckets), point-sized triangles, three
    coordinate scales.  The threaded in-place builder must produce the recursive restatement's node vector byte for
    byte, or fail where it fails (bvh.rs:242)."""
    f32 = np.float32
    mats = np.array([abi.default_material()], dtype=abi.MATERIAL)
    built = errors = 0
    for seed in range(60):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.choice([1, 2, 3, 4, 5, 9, 33, 257, 1000, 4097]))
        scale = f32(rng.choice([1e-4, 1.0, 1e4]))
        c = rng.uniform(-1, 1, size=(n, 3)).astype(f32)
        if rng.random() < 0.3:
            c[:, int(rng.integers(0, 3))] = 0
        if rng.random() < 0.2:
            c[:] = c[0]
        if rng.random() < 0.3:
            c = np.round(c * 4) / 4
        e1, e2 = (rng.uniform(-0.2, 0.2, size=(n, 3)).astype(f32) for _ in range(2))
        k = rng.integers(0, 5, size=n)
        e1[k == 0] = 0
        e2[k == 0] = 0
        t = np.zeros(n, dtype=abi.TRIANGLE)
        t["a"], t["b"], t["c"] = c * scale, (c + e1) * scale, (c + e2) * scale
        arr = abi.SceneArrays(t, None, mats, core.sky_d65(0.0))
        try:
            nodes = core.bvh_build(t)
        except core.PtError as e:
            assert e.code == abi.PT_ERR_BVH_DEGENERATE
            with pytest.raises(Exception):
                oracle.scene(arr)
            errors += 1
            continue
        assert nodes.tobytes() == oracle.scene(arr).nodes().tobytes(), f"seed {seed}, {n} triangles"
        built += 1
    assert built >= 40


def test_allocation_failure_is_an_error_code_not_an_unwind():
    """include/ptcore.h: nothing throws or aborts across the boundary.  A child process caps its address space just above
    what it already holds and calls pt_bvh_build on 2 M triangles: the builder's own vectors (about 90 MB) cannot be had,
    std::bad_alloc is caught at the entry point (csrc/guard.h) and PT_ERR_NOMEM comes back; the process lives on and a small
    build still works afterwards."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, resource, sys
import numpy as np
sys.path.insert(0, %r)
from pathtracer_amd import abi, core
n = 2_000_000
tris = core.gen_soup(1, n, (-1, -1, -4), (1, 1, -2), 0.02, 0.0, 8)
nodes = np.zeros(2 * n - 1, dtype=abi.BVH_NODE)
small = core.gen_soup(2, 1000, (-1, -1, -4), (1, 1, -2), 0.02, 0.0, 8)
vm = int(open("/proc/self/statm").read().split()[0]) * resource.getpagesize()
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
resource.setrlimit(resource.RLIMIT_AS, (vm + (24 << 20), hard))
cnt = C.c_uint32(0)
rc = core.lib().pt_bvh_build(abi.ptr(tris), n, abi.ptr(nodes), C.byref(cnt))
print("rc", rc)
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
print("after", core.bvh_build(small).size)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert f"rc {abi.PT_ERR_NOMEM}" in p.stdout and "after 1999" in p.stdout, p.stdout + p.stderr
