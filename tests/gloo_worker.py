"""One rank of tests/test_sharding.py::test_gloo_world_size_2_sample_split (CPU, gloo).

Usage: RANK/WORLD_SIZE/MASTER_* in the env; argv = spp_total out_dir.
The per-rank renderer is the CPU oracle (test stand-in for pt_render_dev); everything else is the
product's sharding code path: sample_range -> PT_RENDER_SUM partial -> all_reduce(SUM) -> finalize.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pathtracer_amd import abi, sharding, workloads  # noqa: E402
from tests.oracle import Oracle  # noqa: E402


def main():
    spp_total, out_dir = int(sys.argv[1]), sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    o.nthreads = 2
    arr = workloads.box_scene()
    sc = o.scene(arr)
    W, H, seed = 40, 30, 21

    def render_sum(begin, count):
        return torch.from_numpy(o.render(sc, W, H, begin, count, spp_total, seed, mode=abi.PT_RENDER_SUM))

    def all_reduce(buf):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    buf = sharding.render_sharded(render_sum, spp_total, rank, world, all_reduce)
    img = sharding.finalize_sums(buf).numpy()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), img)
    if rank == 0:
        np.save(os.path.join(out_dir, "full.npy"), o.render(sc, W, H, 0, spp_total, spp_total, seed))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
