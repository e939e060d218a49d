"""The library itself on the CPU: tests/cpp/libptcore_emu.so is core.hip, trace.hip, trace_noalloc.hip, render.hip, tonemap.hip and
lbvh.hip - the host logic (scene upload, chunked traces, the pass / slot / look-ahead scheduling of the integrator) and every kernel -
compiled as C++ behind tests/cpp/emu_rt.h: HIP's host API as host memory and synchronous calls, kernels as fibers with the wave
collectives resolved by a scheduler.  The checks (tests/emu_lib_checks.py) drive it through the SAME ctypes binding as the real
library, in a child process (the binding picks its library at import), and hold it to the GPU suite's bar: hits and images
bit-identical to the oracle's.  The last test repeats them in a build with the host's address and undefined-behaviour
sanitizers, which see every "device" buffer, LDS array and workspace carve-out, every shift and every index.  (hipCUB's sort and scan, which lbvh.hip calls, are std:: algorithms there:
tests/cpp/emu_include.)"""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHECKS = ["trace", "trace_dev_and_noalloc", "trace_submit_wait", "loose_tree", "lbvh_and_node_paths", "render_skies", "render_passes_and_sums", "scene_file", "tonemap", "errors"]


def run_checks(lib, checks, preload=None, timeout=1500, shuffle=False):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), lib])
    # (PTC_FLATTEN / PTC_WIDE_COLLAPSE = 2: every scene is flattened and collapsed by the kernels AND by the host loops, and
    # creation fails unless the two agree on every byte; PTC_HOST_CHUNK: pt_trace_closest cuts its batches into several chunks, whose
    # launches alternate between two trace slots)
    env = dict(os.environ, PTCORE_LIB=os.path.join(HERE, "cpp", lib), PTC_FLATTEN="2", PTC_WIDE_COLLAPSE="2", PTC_PASS_PATHS="65536", PTC_HOST_CHUNK="4096",
               PYTHONPATH=ROOT)
    if shuffle:          # the blocks of every launch in a random order (the GPU's dispatch order is undefined)
        env["PTC_EMU_SHUFFLE"], env["PTC_EMU_SEED"] = "1", "7"
    if preload:
        env["LD_PRELOAD"] = (preload + " " + os.environ.get("LD_PRELOAD", "")).strip()
        env["ASAN_OPTIONS"] = "detect_leaks=0"
        env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    p = subprocess.run([sys.executable, "-m", "tests.emu_lib_checks"] + checks, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0 and "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[-3000:] + p.stdout[-500:]
    for c in checks:
        assert f"ok {c}" in p.stdout, (c, p.stdout[-500:])


@pytest.mark.parametrize("check", CHECKS)
def test_the_library_compiled_for_the_cpu(check):
    run_checks("libptcore_emu.so", [check], shuffle=True)


def emu_env(tmp_path):
    """an environment in which programs linked against libptcore.so find the CPU build under that name"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), "libptcore_emu.so"])
    d = tmp_path / "lib"
    d.mkdir()
    os.symlink(os.path.join(HERE, "cpp", "libptcore_emu.so"), d / "libptcore.so")
    return dict(os.environ, LD_LIBRARY_PATH=str(d) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))


def test_cpp_host_program_against_the_cpu_build(tmp_path):
    """tests/cpp/test_ptcore --all (the reference's triangle and camera tests over the kernels, BVH::build / traverse / render /
    Scene::load through include/ptcore.hpp against the oracle): the program the GPU suite runs, unchanged, with the CPU build
    of the library in libptcore.so's place."""
    exe = os.path.join(HERE, "cpp", "test_ptcore")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "test_ptcore"])
    p = subprocess.run([exe, "--all", str(tmp_path)], env=emu_env(tmp_path), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "all checks passed" in p.stdout and "scene::load" in p.stdout, p.stdout + p.stderr


def test_ptrender_checkpoint_and_resume_against_the_cpu_build(tmp_path):
    """harness/ptrender on a 40 x 40 copy of examples/cornell: a render stopped after 2 of 6 samples and resumed from its
    checkpoint is the uninterrupted chunked render bit for bit, and the one-call render's PFM equals the library's image."""
    import shutil
    import numpy as np
    exe = os.path.join(ROOT, "harness", "ptrender")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "harness")])
    scene_dir = tmp_path / "cornell"
    shutil.copytree(os.path.join(ROOT, "examples", "cornell"), scene_dir)
    toml = (scene_dir / "scene.toml").read_text().replace("resolution = [512, 512]", "resolution = [40, 40]")
    assert "[40, 40]" in toml
    (scene_dir / "scene.toml").write_text(toml)
    scene = str(scene_dir / "scene.toml")
    env = emu_env(tmp_path)
    a, b = str(tmp_path / "whole"), str(tmp_path / "resumed")
    common = ["--spp", "6", "--seed", "4", "--chunk", "2"]
    p = subprocess.run([exe, scene, a] + common, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    p = subprocess.run([exe, scene, b] + common + ["--stop-after", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 3, p.stdout + p.stderr
    p = subprocess.run([exe, scene, b] + common + ["--resume"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert open(a + ".pfm", "rb").read() == open(b + ".pfm", "rb").read()


def test_under_the_address_and_undefined_behaviour_sanitizers():
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no address sanitizer runtime for the ROCm clang")
    run_checks("libptcore_emu_asan.so", [c for c in CHECKS if c != "render_passes_and_sums"], preload=rt[0])
