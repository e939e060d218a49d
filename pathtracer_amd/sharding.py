"""Sample-split multi-GPU rendering.

The scene is replicated on every rank, rank g renders a contiguous range of sample indices of
every pixel into a PT_RENDER_SUM buffer (raw XYZ sums, w = samples in the range) and the buffers
are summed with ONE collective — `torch.distributed.all_reduce` / `reduce`, i.e. RCCL over xGMI
under the "nccl" backend, gloo in the CPU tests.  Because sample s of pixel p always draws from
stream (seed, p, s), the result does not depend on the number of ranks beyond f32 summation order.
"""
import numpy as np


def sample_range(spp_total, rank, world_size):
    """Contiguous [begin, begin+count) slice of sample indices for `rank` (balanced to +-1)."""
    if spp_total < 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad sharding arguments")
    base, rem = divmod(spp_total, world_size)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def finalize_sums(sums):
    """PT_RENDER_SUM buffer(s) -> the reference's image: xyz * (1/samples), alpha = 1 (main.rs:49,100).

    Works on numpy arrays and torch tensors of shape [..., 4]."""
    w = sums[..., 3:4]
    inv = 1.0 / w          # w == total samples for every pixel
    out = sums * inv
    out[..., 3] = 1.0
    return out


def render_sharded(render_sum, spp_total, rank, world_size, all_reduce=None):
    """render_sum(begin, count) -> SUM buffer for that range (device tensor or ndarray).

    all_reduce(buf) must sum `buf` over ranks in place (None when world_size == 1).  Returns the
    summed buffer (not yet finalised) on every rank."""
    begin, count = sample_range(spp_total, rank, world_size)
    buf = render_sum(begin, count)
    if world_size > 1 and all_reduce is None:
        raise ValueError("all_reduce is required when world_size > 1")
    if all_reduce is not None:
        all_reduce(buf)
    return buf
