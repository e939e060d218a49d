"""pt_trace_closest_dev_noalloc (include/ptcore.h): the closest-hit entry point that allocates nothing - its stack spill
columns, ticket words and redo list live in a fixed file-scope device array (trace_noalloc.hip) instead of the context's
grown-on-demand buffers.  Called through the C ABI, its output must equal pt_trace_closest_dev's bit for bit: on two calls
in a row, the second with a smaller input (the array keeps the first call's contents), in every traversal mode, with and
without start triangles; sizes beyond the stated maximum are refused with a clear error.

Collected last on purpose (the round's end runs pytest -x): passes alone in a fresh process and after the other tests."""
import numpy as np
import pytest

from pathtracer_amd import abi, core, workloads

pytestmark = pytest.mark.gpu


def _dev(torch, a):
    return torch.from_numpy(a.view(np.uint8).reshape(a.size, -1).copy()).cuda()


def test_noalloc_entry_point_equals_the_allocating_one(gpu_ctx):
    import torch
    lib = core.lib()
    arr = workloads.soup_scene(100_000)
    sc = gpu_ctx.scene(arr)
    rays = core.gen_primary_rays(640, 360)
    rays["d"][5] = 0.0                                   # irregular-ish rays: zero direction, infinite origin (redo path)
    rays["o"][6, 0] = np.inf
    n = rays.size
    rng = np.random.default_rng(1)
    start = np.where(rng.random(n) < 0.3, rng.integers(0, arr.triangles.size, n), abi.PT_NO_TRIANGLE).astype(np.uint32)
    d_rays, d_start = _dev(torch, rays), torch.from_numpy(start.view(np.int32)).cuda()
    want = torch.zeros((n, 16), dtype=torch.uint8, device="cuda")
    got = torch.zeros((n, 16), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    try:
        for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_MIN_T, abi.PT_TRAVERSE_REPLAY):
            gpu_ctx.set_traversal(mode)
            # two calls in a row, the second smaller and with start triangles
            for m, st in ((n, 0), (n // 3, d_start.data_ptr()), (257, 0)):
                want.zero_(); got.zero_()
                torch.cuda.synchronize()
                gpu_ctx.trace_dev(sc, d_rays.data_ptr(), st, m, want.data_ptr())
                gpu_ctx.trace_dev_noalloc(sc, d_rays.data_ptr(), st, m, got.data_ptr())
                torch.cuda.synchronize()
                assert gpu_ctx.stats().trace_rays == m
                assert torch.equal(got, want), (mode, m)
                assert int((want[:m, :4].view(torch.int32) != -1).sum()) > m // 2          # it did trace something
                assert not got[m:].any()                                                   # and wrote nothing past its batch
    finally:
        gpu_ctx.set_traversal(abi.PT_TRAVERSE_DEFAULT)
    # n = 0 is fine; beyond the maximum is an error code with a message, not a crash
    assert lib.pt_trace_closest_dev_noalloc(gpu_ctx.handle, sc.handle, None, None, 0, None) == abi.PT_OK
    rc = lib.pt_trace_closest_dev_noalloc(gpu_ctx.handle, sc.handle, core.C.c_void_p(d_rays.data_ptr()), None,
                                          abi.PT_NOALLOC_MAX_RAYS + 1, core.C.c_void_p(got.data_ptr()))
    assert rc == abi.PT_ERR_UNSUPPORTED and b"PT_NOALLOC_MAX_RAYS" in lib.pt_last_error(gpu_ctx.handle)
