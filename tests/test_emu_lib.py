"""The library itself on the CPU: tests/cpp/libptcore_emu.so is core.hip, trace.hip, trace_noalloc.hip, render.hip, tonemap.hip and
lbvh.hip - the host logic (scene upload, chunked traces, the pass / slot / look-ahead scheduling of the integrator) and every kernel -
compiled as C++ behind tests/cpp/emu_rt.h: HIP's host API as host memory and synchronous calls, kernels as fibers with the wave
collectives resolved by a scheduler.  The checks (tests/emu_lib_checks.py) drive it through the SAME ctypes binding as the real
library, in a child process (the binding picks its library at import), and hold it to the GPU suite's bar: hits and images
bit-identical to the oracle's.  The last test repeats them under the host's address sanitizer, which sees every "device"
buffer, LDS array and workspace carve-out.  (hipCUB's sort and scan, which lbvh.hip calls, are std:: algorithms there:
tests/cpp/emu_include.)"""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHECKS = ["trace", "trace_dev_and_noalloc", "loose_tree", "lbvh_and_node_paths", "render_skies", "render_passes_and_sums", "scene_file", "tonemap", "errors"]


def run_checks(lib, checks, preload=None, timeout=1500):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), lib])
    # (PTC_FLATTEN / PTC_WIDE_COLLAPSE = 2: every scene is flattened and collapsed by the kernels AND by the host loops, and
    # creation fails unless the two agree on every byte)
    env = dict(os.environ, PTCORE_LIB=os.path.join(HERE, "cpp", lib), PTC_FLATTEN="2", PTC_WIDE_COLLAPSE="2", PTC_PASS_PATHS="65536",
               PYTHONPATH=ROOT)
    if preload:
        env["LD_PRELOAD"] = (preload + " " + os.environ.get("LD_PRELOAD", "")).strip()
        env["ASAN_OPTIONS"] = "detect_leaks=0"
    p = subprocess.run([sys.executable, "-m", "tests.emu_lib_checks"] + checks, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0 and "ERROR: AddressSanitizer" not in p.stderr, p.stderr[-3000:] + p.stdout[-500:]
    for c in checks:
        assert f"ok {c}" in p.stdout, (c, p.stdout[-500:])


@pytest.mark.parametrize("check", CHECKS)
def test_the_library_compiled_for_the_cpu(check):
    run_checks("libptcore_emu.so", [check])


def test_under_the_address_sanitizer():
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no address sanitizer runtime for the ROCm clang")
    run_checks("libptcore_emu_asan.so", [c for c in CHECKS if c != "render_passes_and_sums"], preload=rt[0])
