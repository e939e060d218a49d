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
