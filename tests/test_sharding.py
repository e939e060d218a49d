"""Multi-GPU sharding logic (pathtracer_amd/sharding.py) on the CPU: sample-range arithmetic, and a
world_size-2 `gloo` run of render_sharded whose per-rank renderer is the CPU oracle standing in
for pt_render (the collective, the range split and the finalisation are what is under test; on the
GPU box the same code runs with backend "nccl" = RCCL and pt_render_dev, see bench.py).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from pathtracer_amd import abi, sharding, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("spp,world", [(1024, 8), (64, 1), (7, 3), (3, 8), (0, 2), (1023, 4)])
def test_sample_range_partitions_the_samples(spp, world):
    seen = []
    for r in range(world):
        b, c = sharding.sample_range(spp, r, world)
        assert c in (spp // world, spp // world + 1)
        seen.extend(range(b, b + c))
    assert seen == list(range(spp))


def test_sample_range_rejects_bad_arguments():
    for args in ((8, 2, 2), (8, -1, 2), (8, 0, 0), (-1, 0, 1)):
        with pytest.raises(ValueError):
            sharding.sample_range(*args)


def test_render_sharded_needs_a_collective_when_sharded():
    with pytest.raises(ValueError):
        sharding.render_sharded(lambda b, c: np.zeros((2, 2, 4), np.float32), 8, 0, 2, None)


def test_finalize_sums():
    s = np.zeros((2, 3, 4), dtype=np.float32)
    s[..., :3] = 6.0
    s[..., 3] = 4.0
    out = sharding.finalize_sums(s.copy())
    assert (out[..., :3] == 1.5).all() and (out[..., 3] == 1.0).all()


def test_partial_sums_compose_on_one_process(oracle):
    """SUM([0,3)) + SUM([3,8)) == SUM([0,8)) up to f32 summation order; FINAL == finalize(SUM)."""
    arr = workloads.box_scene()
    sc = oracle.scene(arr)
    a = oracle.render(sc, 48, 32, 0, 3, 8, seed=5, mode=abi.PT_RENDER_SUM)
    b = oracle.render(sc, 48, 32, 3, 5, 8, seed=5, mode=abi.PT_RENDER_SUM)
    full = oracle.render(sc, 48, 32, 0, 8, 8, seed=5, mode=abi.PT_RENDER_SUM)
    np.testing.assert_allclose(a + b, full, rtol=2e-6, atol=1e-9)
    final = oracle.render(sc, 48, 32, 0, 8, 8, seed=5)
    np.testing.assert_allclose(sharding.finalize_sums(full.copy()), final, rtol=2e-7, atol=0)
    assert (final[..., 3] == 1.0).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,spp", [(2, 6), (2, 5)])
def test_gloo_world_size_2_sample_split(tmp_path, world, spp):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gloo_worker.py"), str(spp), str(tmp_path)],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
    imgs = [np.load(os.path.join(tmp_path, f"rank{r}.npy")) for r in range(world)]
    # every rank holds the same reduced image ...
    assert np.array_equal(imgs[0], imgs[1])
    # ... and it equals the unsharded render up to f32 summation order
    full = np.load(os.path.join(tmp_path, "full.npy"))
    np.testing.assert_allclose(imgs[0], full, rtol=2e-6, atol=1e-9)
    assert (imgs[0][..., 3] == 1.0).all() and imgs[0][..., 1].mean() > 0
