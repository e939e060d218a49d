"""The C++ host side (include/ptcore.hpp) through its own test program, tests/cpp/test_ptcore.cpp: the
reference's triangle and camera tests restated over the GPU path, BVH::build / traverse / render / Scene::load
against the CPU oracle.  The program links libptcore.so against the SYSTEM ROCm runtime (no PyTorch in the
process), which is how a Rust or C++ host would run it."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "cpp", "test_ptcore")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):          # built by __graft_entry__.build(); here only as a convenience on a dev box
        subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp")])
    return EXE


def run(args):
    p = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "all checks passed" in p.stdout, p.stdout + p.stderr
    return p.stdout


def test_header_is_plain_cxx17(tmp_path):
    """ptcore.hpp needs nothing but ptcore.h and the standard library (no HIP, no torch)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "ptcore.hpp"\nint main() { return sizeof(pathtracer::Scene) == 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(HERE, "..", "include"), str(src)])


def test_host_cases(exe):
    out = run([exe, "--host"])
    assert "bvh::build" in out and "camera::corners" in out


@pytest.mark.gpu
def test_gpu_cases(exe, tmp_path):
    out = run([exe, "--all", str(tmp_path)])
    assert "triangle::intersection" in out and "scene::load" in out


@pytest.mark.gpu
def test_ptrender_harness(gpu_ctx, tmp_path):
    """harness/ptrender: scene file in, PNG + PFM out, image equal to what the Python harness renders from the same file."""
    import numpy as np
    from pathtracer_amd import core
    root = os.path.dirname(HERE)
    exe = os.path.join(root, "harness", "ptrender")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "harness")])
    scene = os.path.join(root, "examples", "cornell", "scene.toml")
    out = str(tmp_path / "img")
    p = subprocess.run([exe, scene, out, "--spp", "4", "--seed", "5"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "rendered 512x512, 4 spp" in p.stdout, p.stdout + p.stderr
    raw = open(out + ".pfm", "rb").read()
    assert raw.startswith(b"PF\n512 512\n-1.0\n")
    pfm = np.frombuffer(raw[len(b"PF\n512 512\n-1.0\n"):], dtype="<f4").reshape(512, 512, 3)[::-1]
    assert open(out + ".png", "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    s = core.load_scene_file(scene)
    img = gpu_ctx.render(gpu_ctx.scene(s.arrays), 512, 512, 0, 4, 4, seed=5, sensor_width=s.camera["sensor_width"], focal_length=s.camera["focal_length"])
    assert np.array_equal(pfm, img[..., :3])
    bad = subprocess.run([exe, str(tmp_path / "absent.toml"), out], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "cannot open" in bad.stderr


@pytest.mark.gpu
def test_ptrender_checkpoint_and_resume(tmp_path):
    """harness/ptrender --chunk K: sample ranges of PT_RENDER_SUM added on the host, a checkpoint (sums + samples done) after every
    range.  A render stopped after 2 of 6 samples and resumed from its checkpoint is the uninterrupted chunked render bit for
    bit (sample s of pixel p draws from stream (seed, p, s) whatever ran before); a checkpoint of another render is refused."""
    root = os.path.dirname(HERE)
    exe = os.path.join(root, "harness", "ptrender")
    scene = os.path.join(root, "examples", "cornell", "scene.toml")
    a, b = str(tmp_path / "whole"), str(tmp_path / "resumed")
    common = ["--spp", "6", "--seed", "4", "--chunk", "2"]
    p = subprocess.run([exe, scene, a] + common, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    q = subprocess.run([exe, scene, b] + common + ["--stop-after", "2"], capture_output=True, text=True, timeout=300)
    assert q.returncode == 3 and "stopped with 2 of 6 samples" in q.stdout and os.path.exists(b + ".ckpt") and not os.path.exists(b + ".pfm")
    wrong = subprocess.run([exe, scene, b, "--spp", "6", "--seed", "5", "--chunk", "2", "--resume"], capture_output=True, text=True, timeout=300)
    assert wrong.returncode == 1 and "belongs to another render" in wrong.stderr
    r = subprocess.run([exe, scene, b] + common + ["--resume"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "resuming at sample 2 of 6" in r.stdout, r.stdout + r.stderr
    assert open(a + ".pfm", "rb").read() == open(b + ".pfm", "rb").read()
    assert open(a + ".ckpt", "rb").read() == open(b + ".ckpt", "rb").read()
    # a damaged checkpoint is refused by its checksum
    raw = bytearray(open(b + ".ckpt", "rb").read())
    raw[-5] ^= 0x40
    open(b + ".ckpt", "wb").write(bytes(raw))
    bad = subprocess.run([exe, scene, b] + common + ["--resume"], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1 and "checksum" in bad.stderr


def test_ptrender_resume_needs_a_chunk_size():
    root = os.path.dirname(HERE)
    exe = os.path.join(root, "harness", "ptrender")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "harness"), "ptrender"])
    p = subprocess.run([exe, "x.toml", "out", "--resume"], capture_output=True, text=True, timeout=30)
    assert p.returncode == 2 and "--chunk" in p.stderr


@pytest.mark.gpu
def test_ptrender_mgpu_single_rank_equals_ptrender(tmp_path):
    """harness/ptrender_mgpu: the sample-split render from a compiled host - rank processes forked before any GPU call,
    ncclCommInitRank, PT_RENDER_SUM into HBM, ncclAllReduce(ncclFloat, ncclSum), finalize.  With one rank (all this box has)
    the image must be ptrender's bit for bit; more ranks than GPUs is an error, not a hang.  Unmeasured at N > 1."""
    import json
    root = os.path.dirname(HERE)
    for exe in ("ptrender", "ptrender_mgpu"):
        if not os.path.exists(os.path.join(root, "harness", exe)):
            subprocess.check_call(["make", "-C", os.path.join(root, "harness")])
    scene = os.path.join(root, "examples", "cornell", "scene.toml")
    a, b = str(tmp_path / "one"), str(tmp_path / "mgpu")
    p = subprocess.run([os.path.join(root, "harness", "ptrender"), scene, a, "--spp", "6", "--seed", "9"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    q = subprocess.run([os.path.join(root, "harness", "ptrender_mgpu"), scene, b, "--gpus", "1", "--spp", "6", "--seed", "9"],
                       capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stdout + q.stderr
    line = json.loads(q.stdout.strip().splitlines()[-1])
    assert line["gpus"] == 1 and line["spp"] == 6 and line["allreduce_bytes"] == 512 * 512 * 16 and line["spp_per_s"] > 0
    assert open(a + ".pfm", "rb").read() == open(b + ".pfm", "rb").read()
    import torch
    if torch.cuda.device_count() == 1:
        r = subprocess.run([os.path.join(root, "harness", "ptrender_mgpu"), scene, b, "--gpus", "2", "--spp", "6"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "one rank per GPU" in r.stderr


@pytest.mark.gpu
def test_ptrender_mgpu_soup_workload_equals_the_python_harness(gpu_ctx, tmp_path):
    """configs 3 / 4 from the compiled host: `--soup N --sky hw --size WxH` builds the same soup (generator, materials, sky) as
    pathtracer_amd/workloads.py, so its image must be the Python harness's bit for bit; a scene that cannot be loaded ends
    the ranks by agreement (exit 1, message), not by a signal."""
    import json
    import numpy as np
    from pathtracer_amd import workloads
    root = os.path.dirname(HERE)
    exe = os.path.join(root, "harness", "ptrender_mgpu")
    out = str(tmp_path / "soup")
    q = subprocess.run([exe, "--soup", "20000", out, "--gpus", "1", "--spp", "3", "--seed", "4", "--size", "320x180", "--sky", "hw"],
                       capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stdout + q.stderr
    line = json.loads(q.stdout.strip().splitlines()[-1])
    assert line["ranks_seen"] == 1 and line["width"] == 320 and line["height"] == 180 and line["spp"] == 3 and "soup-20000" in line["workload"]
    raw = open(out + ".pfm", "rb").read()
    head = b"PF\n320 180\n-1.0\n"
    assert raw.startswith(head)
    pfm = np.frombuffer(raw[len(head):], dtype="<f4").reshape(180, 320, 3)[::-1]
    arr = workloads.soup_scene(20000, sky=workloads.hw_sky())
    img = gpu_ctx.render(gpu_ctx.scene(arr), 320, 180, 0, 3, 3, seed=4)
    got, want = pfm, img[..., :3]
    assert ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all()
    bad = subprocess.run([exe, str(tmp_path / "absent.toml"), out, "--gpus", "1"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "could not load the scene" in bad.stderr


def test_ptrender_mgpu_usage():
    root = os.path.dirname(HERE)
    exe = os.path.join(root, "harness", "ptrender_mgpu")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "harness"), "ptrender_mgpu"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=30)
    assert p.returncode == 2 and "usage" in p.stderr
