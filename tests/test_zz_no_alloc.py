"""pt_trace_closest_dev_noalloc (include/ptcore.h): the closest-hit entry point that allocates nothing - its stack spill
columns, ticket words and redo list live in a fixed file-scope device array (trace_noalloc.hip) instead of the context's
grown-on-demand buffers.  Called through the C ABI, its output must equal pt_trace_closest_dev's bit for bit: on two calls
in a row, the second with a smaller input (the array keeps the first call's contents), in every traversal mode, with and
without start triangles; sizes beyond the stated maximum are refused with a clear error; and the free device memory does not
move across its calls on a context that has never run the allocating entry point.

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


def _chain(n):
    from tests.test_gpu_trace import chain_scene
    return chain_scene(n)


def test_noalloc_allocates_nothing_and_knows_its_limits(oracle):
    """A FRESH context that only ever runs the no-alloc entry point (so none of the context's grown-on-demand buffers exist
    and an accidental ptc_reserve could not hide behind an earlier call): after one small warm-up call (code objects, kernel
    argument pools) the free device memory is the same before and after larger calls in every mode.  A 599-level chain tree
    takes the reduced-grid path (its stack columns are deep, fewer blocks fit the fixed array); a 4095-level chain is
    refused for PT_TRAVERSE_EXACT with a message and still answered by REPLAY, whose columns are half as deep."""
    import torch
    lib = core.lib()
    ctx = core.Context(0)
    try:
        arr, deep, too_deep = workloads.soup_scene(50_000), _chain(600), _chain(4096)
        sc, sc_deep, sc_too_deep = ctx.scene(arr), ctx.scene(deep), ctx.scene(too_deep)
        assert sc_deep.info().max_depth == 599 and sc_too_deep.info().max_depth == 4095
        rays = core.gen_primary_rays(640, 360)
        n = rays.size
        d_rays = _dev(torch, rays)
        got = torch.zeros((n, 16), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.set_traversal(abi.PT_TRAVERSE_MIN_T)
        ctx.trace_dev_noalloc(sc, d_rays.data_ptr(), 0, 257, got.data_ptr())          # warm-up: small, the cheapest mode
        ctx.set_traversal(abi.PT_TRAVERSE_REPLAY)
        ctx.trace_dev_noalloc(sc, d_rays.data_ptr(), 0, 257, got.data_ptr())
        ctx.set_traversal(abi.PT_TRAVERSE_EXACT)
        ctx.trace_dev_noalloc(sc, d_rays.data_ptr(), 0, 257, got.data_ptr())
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        results = {}
        for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_MIN_T, abi.PT_TRAVERSE_REPLAY):
            ctx.set_traversal(mode)
            for name, s in (("soup", sc), ("deep", sc_deep)):
                ctx.trace_dev_noalloc(s, d_rays.data_ptr(), 0, n, got.data_ptr())
                torch.cuda.synchronize()
                results[(mode, name)] = got.cpu().numpy().copy()
        # too deep for the fixed array in EXACT (4 entries per wide level: 8 196 x 2 KiB per block, fewer than 8 blocks fit) ...
        ctx.set_traversal(abi.PT_TRAVERSE_EXACT)
        rc = lib.pt_trace_closest_dev_noalloc(ctx.handle, sc_too_deep.handle, core.C.c_void_p(d_rays.data_ptr()), None, 4096,
                                              core.C.c_void_p(got.data_ptr()))
        assert rc == abi.PT_ERR_UNSUPPORTED and b"too deep" in lib.pt_last_error(ctx.handle)
        # ... and fine for REPLAY (one frame per binary level)
        ctx.set_traversal(abi.PT_TRAVERSE_REPLAY)
        ctx.trace_dev_noalloc(sc_too_deep, d_rays.data_ptr(), 0, 4096, got.data_ptr())
        torch.cuda.synchronize()
        too_deep_hits = got[:4096].cpu().numpy().copy()
        free_after = torch.cuda.mem_get_info()[0]
        assert free_after == free_before, (free_before, free_after)
        # what those calls returned is the oracle's answer (the deep trees too: capped grid, long stack columns)
        for name, a in (("soup", arr), ("deep", deep)):
            want = oracle.trace(oracle.scene(a), rays)
            for mode in (abi.PT_TRAVERSE_EXACT, abi.PT_TRAVERSE_REPLAY):
                assert results[(mode, name)].tobytes() == want.tobytes(), (mode, name)
            assert np.array_equal(results[(abi.PT_TRAVERSE_MIN_T, name)].view(np.uint32).reshape(-1, 4)[:, 1], want["t"].view(np.uint32)), name
        want = oracle.trace(oracle.scene(too_deep), rays[:4096])
        assert too_deep_hits.tobytes() == want.tobytes()
    finally:
        ctx.close()
