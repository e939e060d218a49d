#!/usr/bin/env python3
"""bench.py — headline benchmark of the render core (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline (BASELINE.json configs[1]): closest-hit of the 2 073 600 pixel-centre camera rays of a
1920x1080 frame against the synthetic 1M-triangle soup, rays and hits resident in HBM, default traversal
(PT_TRAVERSE_EXACT: the reference's answer on every ray).  One STEP = one pt_trace_closest_dev call.  The path shards with
no exchange; with N > 1
  --scaling weak   (default) every rank holds the scene and traces its own full-size batch (rank r shifts the pixel grid by
                   r/N of a pixel so the rays differ): value = all rays / time;
  --scaling strong the ONE batch is cut into N contiguous chunks, rank r traces chunk r: value = the batch's
                   rays / the slowest rank's time.

The same JSON line also carries
  roofline      the ceiling that binds the traversal kernel (VALU issue or the vector L1's lane-load rate, both measured on
                this GPU model by harness/ptissue and harness/ptgather; instruction and access counts from the rocprofv3
                PMC passes committed under profiles/, time measured live with HIP events on the library's stream), and
                beside it the HBM figures SURVEY §8(d) defines: hbm_algorithmic_frac (B_ray = 32 + 16 + 64 n_node + 48 n_tri
                under the REFERENCE's traversal rule, counted by the CPU oracle: re-reads served by L1/L2/Infinity Cache make
                it exceed 1) and hbm_counter_frac (memory-side bytes from the counters);
  cpu_baseline  the oracle (C++ restatement of the reference algorithm) on this box's host cores;
  pipelined     the same K steps submitted without a wait between them (pt_trace_closest_dev_submit, --in-flight 2 batches on
                streams of their own): what a caller that has the next sample's batch ready gets; not the headline;
  spectral      BASELINE configs[2]: spectral path tracing, soup-1M + Hosek-Wilkie sky, 1080p, 64 spp per rank (weak), with
                its own roofline (algorithmic bytes per path-sample) and cpu_baseline;
  config4       BASELINE configs[3]: the same scene at 3840x2160, --config4-spp samples IN TOTAL (default 1024) split across
                the ranks (strong scaling), PT_RENDER_SUM buffers combined with ONE RCCL all-reduce of 132.7 MB, timed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


KERNELS = {"exact": "k_trace_scoped_persistent", "min_t": "k_trace_mint_persistent", "replay": "k_trace_replay_persistent"}


def host_threads():
    """Threads the CPU baseline runs on: the CPUs this process may use, at most the box's per-GPU share of 16."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launcher_command(argv, gpus, port):
    """The command `python bench.py --gpus N` (N > 1, no RANK in the environment) starts as a CHILD: one rank per GPU under
    torch.distributed.run, the same arguments handed through."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(a, argv):
    """Called before torch is imported and before anything touches the GPU.  Starts the ranks as a child process (never an
    exec), relays rank 0's JSON line and everything else the ranks print, returns the child's exit status."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")
    cmd = launcher_command(argv, a.gpus, free_port())
    print("bench.py: starting %d ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    seen_json = False
    for line in child.stdout:              # relayed as it arrives: a long run keeps printing
        if line.lstrip().startswith("{") and ('"metric"' in line or '"launch_check"' in line):
            seen_json = True
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = child.wait()
    if rc == 0 and not seen_json:
        print("bench.py: the ranks ended without a JSON line", file=sys.stderr)
        rc = 1
    return rc


def launch_check(a):
    """--launch-check: the rendezvous of the N ranks and nothing else (no GPU call when the backend is gloo): an all-reduce of
    ones proves how many ranks the process group holds.  What tests/test_bench_contract.py runs on the CPU."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if a.backend == "nccl":
        local_rank = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        ones = torch.ones(1, device="cuda")
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
        ones = torch.ones(1)
    dist.all_reduce(ones)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": a.gpus, "world_size": world, "ranks_seen": int(ones.item()),
                          "backend": dist.get_backend()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="closest-hit leg with N > 1 (see the module docstring)")
    ap.add_argument("--config4-spp", type=int, default=1024, help="samples per pixel IN TOTAL of the 3840x2160 leg, split across ranks (config 4: 1024; 0 = skip)")
    ap.add_argument("--spectral-spp", type=int, default=64, help="samples per pixel PER RANK of the spectral leg (config 3: 64; 0 = skip)")
    ap.add_argument("--traversal", choices=["exact", "min_t", "replay"], default="exact")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight of the `pipelined` leg (pt_trace_closest_dev_submit; 1 = skip the leg, as the profiling scripts do)")
    ap.add_argument("--backend", default="nccl", help="process-group backend; 'gloo' + --one-device rehearses the N>1 code path on a 1-GPU box")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, count them with one all-reduce, print that and stop (no benchmark)")
    a = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 starts its own ranks: a child `python -m torch.distributed.run` BEFORE torch is
    # imported or the GPU touched in this process (under torch.distributed.run RANK is set and this is a rank already)
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(a, sys.argv[1:]))
    if a.launch_check:
        return launch_check(a)

    import torch
    import torch.distributed as dist
    from pathtracer_amd import abi, core, workloads, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} but WORLD_SIZE is {world}: one rank per GPU")
    if a.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is created even for one rank, so that the
    # RCCL path of this script can be exercised on a single GPU
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    # how many ranks the process group really holds: an all-reduce of ones over the backend the legs below use
    ranks_seen = 1
    if use_dist:
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(*vals):
        if not use_dist:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(x) for x in t)

    # ---- scene + rays (host), upload ---------------------------------------------------------
    t0 = time.time()
    arr = workloads.soup_scene(a.tris, sky=workloads.hw_sky())
    build_s = time.time() - t0
    ctx = core.Context(local_rank)
    scene = ctx.scene(arr)
    ctx.set_traversal({"exact": abi.PT_TRAVERSE_EXACT, "min_t": abi.PT_TRAVERSE_MIN_T, "replay": abi.PT_TRAVERSE_REPLAY}[a.traversal])
    info = scene.info()
    rays = core.gen_primary_rays(a.width, a.height)
    batch = rays.size
    strong = world > 1 and a.scaling == "strong"
    if strong:
        lo, hi = batch * rank // world, batch * (rank + 1) // world      # rank r's contiguous chunk of the ONE batch
        rays = np.ascontiguousarray(rays[lo:hi])
    elif world > 1 and rank > 0:
        # weak: distinct rays per rank, the pixel grid shifted by rank/world of a pixel (camera.rs:19-28)
        cam_sx = np.float32(36.0) / np.float32(a.width + 1)
        rays["d"][:, 0] += np.float32(rank / world) * cam_sx
    n = rays.size
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(n, 8)).cuda()
    d_hits = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()        # the library runs on its own stream: torch's fills must be complete first

    # ---- timed region --------------------------------------------------------------------------
    for _ in range(a.warmup):
        ctx.trace_dev(scene, d_rays.data_ptr(), 0, n, d_hits.data_ptr())
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(a.steps):
        ctx.trace_dev(scene, d_rays.data_ptr(), 0, n, d_hits.data_ptr())
        kernel_ms += ctx.stats().trace_ms          # HIP events around the kernel(s), on the library's own stream
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, kernel_ms = max_over_ranks(elapsed, kernel_ms)
    total_rays = (batch if strong else n * world) * a.steps
    mrays = total_rays / elapsed / 1e6
    kernel_ms_avg = kernel_ms / a.steps
    # ---- the same K steps without a wait between them (pt_trace_closest_dev_submit / _wait): --in-flight batches queued on
    # slots of their own, so that the drain of one launch fills with the head of the next.  Reported beside the headline,
    # which stays the blocking call (one launch alone on the chip: what `roofline` and the rocprofv3 summaries describe).
    pipelined = None
    if a.in_flight > 1:
        try:
            F = min(a.in_flight, abi.PT_TRACE_SLOTS)
            d_hits_f = [torch.zeros_like(d_hits) for _ in range(F)]
            torch.cuda.synchronize()

            def run(steps):
                for i in range(steps):
                    slot = i % F
                    if i >= F:
                        ctx.trace_dev_wait(slot)              # the slot's hits are complete: its buffer is free again
                    ctx.trace_dev_submit(scene, d_rays.data_ptr(), 0, n, d_hits_f[slot].data_ptr(), slot)
                for slot in range(F):
                    ctx.trace_dev_wait(slot)

            run(max(a.warmup, F))
            barrier()
            t0 = time.perf_counter()
            run(a.steps)
            barrier()
            (p_elapsed,) = max_over_ranks(time.perf_counter() - t0)
            same = all(bool(torch.equal(h, d_hits)) for h in d_hits_f[:min(F, a.steps)])
            pipelined = {
                "value": (batch if strong else n * world) * a.steps / p_elapsed / 1e6, "unit": "Mrays/s", "in_flight": F,
                "ms_per_step": p_elapsed / a.steps * 1e3, "steps": a.steps, "hits_equal_blocking_call": same,
                "entry_points": "pt_trace_closest_dev_submit / pt_trace_closest_dev_wait",
                "note": "the same K batches, the next one submitted before the previous one's slowest rays have finished "
                        "(consecutive samples' batches are independent, main.rs:64-76); throughput of a caller that submits ahead, "
                        "not the headline",
            }
        except Exception as e:
            pipelined = {"value": None, "unit": "Mrays/s", "error": f"{type(e).__name__}: {e}"}
    # the opt-in global-best traversal on the same batch, for the record (not the headline)
    alt = None
    if a.traversal == "exact":
        ctx.set_traversal(abi.PT_TRAVERSE_MIN_T)
        ms = []
        for _ in range(7):
            ctx.trace_dev(scene, d_rays.data_ptr(), 0, n, d_hits.data_ptr())
            ms.append(ctx.stats().trace_ms)
        alt_ids = d_hits[:, 0].clone()
        ctx.set_traversal(abi.PT_TRAVERSE_EXACT)
        ctx.trace_dev(scene, d_rays.data_ptr(), 0, n, d_hits.data_ptr())
        alt = {"traversal": "min_t", "kernel": KERNELS["min_t"], "kernel_ms": float(np.mean(ms[2:])),
               "Mrays_per_s_per_gpu": n / float(np.mean(ms[2:])) / 1e3, "ids_differing_from_exact": int((alt_ids != d_hits[:, 0]).sum()),
               "note": "PT_TRAVERSE_MIN_T prunes against the best hit of the whole ray: a minimum-t hit, not the recursion's choice at ties (opt-in)"}

    def render_leg(width, height, spp_total, begin, count, label, mode=abi.PT_RENDER_SUM):
        """One sample-split render: this rank renders [begin, begin+count) into a SUM buffer, one all-reduce, finalize.  Timed
        between barriers, max over ranks."""
        d_img = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        reduce_ms = [0.0]

        def render_sum(b, c):
            ctx.render_dev(scene, width, height, b, c, spp_total, 1, mode, d_img.data_ptr())
            return d_img

        def all_reduce(buf):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)       # RCCL over xGMI
            torch.cuda.synchronize()
            reduce_ms[0] = (time.perf_counter() - t1) * 1e3

        if use_dist:   # warm the communicator on a buffer of this size
            dist.all_reduce(torch.zeros_like(d_img))
        # warm-up: one pass-sized call, so that the pass workspace (about 1.15 KiB per path in flight) is allocated before
        # the timed region
        ctx.render_dev(scene, width, height, begin, min(count, max(1, (128 << 20) // (width * height))), spp_total, 1, mode, d_img.data_ptr())
        sharding.finalize_sums(d_img.clone())       # first use of these torch kernels loads their code objects
        barrier()
        t1 = time.perf_counter()
        buf = sharding.render_sharded(lambda b, c: render_sum(begin, count), spp_total, rank, world, all_reduce if use_dist else None)
        img = sharding.finalize_sums(buf)
        barrier()
        sec = time.perf_counter() - t1
        st = ctx.stats()
        (sec,) = max_over_ranks(sec)
        return {
            "metric": label, "value": spp_total / sec, "unit": "spp/s", "spp_total": spp_total, "spp_this_rank": count,
            "seconds": sec, "segments_per_path": st.segments / max(st.paths, 1),
            "trace_Mrays_per_s_rank0": st.trace_rays / max(st.trace_ms, 1e-9) / 1e3,
            "trace_share_of_device_time_rank0": st.trace_ms / max(st.total_ms, 1e-9),
            "trace_launches_rank0": st.trace_launches,
            "device_ms_by_kernel_rank0": {"total": st.total_ms, "trace": st.trace_ms, "shade": st.shade_ms, "sky": st.sky_ms,
                                          "accumulate": st.accumulate_ms, "generate": st.generate_ms},
            "allreduce_ms": reduce_ms[0] if use_dist else None, "allreduce_bytes": int(d_img.numel() * 4),
            "finite_pixel_fraction": float(torch.isfinite(img[..., 1]).float().mean()),
        }, d_img

    # ---- spectral leg (config 3; weak: 64 spp per rank, samples split + all-reduce when N > 1) --------------------
    spectral, d_img = None, None
    if a.spectral_spp > 0:
        try:
            spp_total = a.spectral_spp * world
            spectral, d_img = render_leg(a.width, a.height, spp_total, rank * a.spectral_spp, a.spectral_spp,
                                         "spectral spp/s at 1080p, 1M tris")
            spectral["scaling"] = "weak"
            # the same leg with the opt-in extension that normalises the direction handed to the sky model: the reference does
            # not (hosek_wilkie.rs:313-316 then makes NaN of most pixels, finite_pixel_fraction above), so "images match" on the
            # faithful leg is largely NaN-mask equality; this one is finite everywhere
            nrm, _ = render_leg(a.width, a.height, spp_total, rank * a.spectral_spp, a.spectral_spp, "spectral spp/s at 1080p, 1M tris, sky direction normalised",
                                mode=abi.PT_RENDER_SUM | abi.PT_RENDER_NORMALIZE_SKY_DIRECTION)
            spectral["normalized_sky_direction"] = {k: nrm[k] for k in ("value", "unit", "seconds", "finite_pixel_fraction", "segments_per_path")}
            spectral["config"] = {"workload": f"soup-{a.tris} + Hosek-Wilkie sky (turbidity 3, albedo 0.3), {a.width}x{a.height}, "
                                              f"{spp_total} spp, seed 1 (BASELINE configs[2]: 64 spp per GPU, samples split across ranks)",
                                  "traversal": a.traversal}
        except Exception as e:      # keep the headline line: the closest-hit measurement above is complete
            spectral = {"metric": "spectral spp/s at 1080p, 1M tris", "value": None, "unit": "spp/s", "error": f"{type(e).__name__}: {e}"}

    # ---- config 4: 3840x2160, a FIXED total of samples split across the ranks (strong scaling) -------------------
    config4 = None
    if a.config4_spp > 0:
        try:
            b4, c4 = sharding.sample_range(a.config4_spp, rank, world)
            config4, _ = render_leg(3840, 2160, a.config4_spp, b4, c4, "spectral spp/s at 3840x2160, 1M tris, sample-split")
            config4["scaling"] = "strong"
            config4["config"] = {"workload": f"soup-{a.tris} + Hosek-Wilkie sky, 3840x2160, {a.config4_spp} spp in total, rank g renders "
                                             f"sample indices [g*S/G, (g+1)*S/G), one all-reduce of the 132.7 MB sum buffer (BASELINE configs[3])",
                                 "traversal": a.traversal}
        except Exception as e:
            config4 = {"metric": "spectral spp/s at 3840x2160, 1M tris, sample-split", "value": None, "unit": "spp/s", "error": f"{type(e).__name__}: {e}"}

    # ---- CPU baseline + visit counts (rank 0 only, also at N > 1: bounded, about 10-30 s of CPU work on rank 0's own
    # rays while the other ranks wait at the last barrier) -------
    cpu_baseline, n_node, n_tri, counts_src = None, None, None, None
    golden = os.path.join(ROOT, "tests", "golden", "soup1m_primary_counts.json")
    if rank == 0 and not a.no_cpu_baseline:
        try:
            from tests.oracle import Oracle
            o = Oracle()
            osc = o.scene(arr)
            o.nthreads = host_threads()
            sample = rays                      # the whole batch: ~5-20 s
            t1 = time.perf_counter()
            want, cnt = o.trace(osc, sample, count=True)
            cpu_s = time.perf_counter() - t1
            n_node, n_tri = cnt[0] / sample.size, cnt[1] / sample.size
            counts_src = "oracle, live"
            got = d_hits.cpu().numpy().view(np.uint32)[:, 0]
            cpu_baseline = {
                "value": sample.size / cpu_s / 1e6, "unit": "Mrays/s", "cores": o.nthreads, "kind": "port",
                "host_cpus_visible": os.cpu_count(),
                "sample": f"all {sample.size} rays of the batch, {cpu_s:.1f} s, reference-algorithm CPU restatement "
                          f"(recursive traversal, bvh.rs:250-338) on {o.nthreads} threads (the CPUs this process may use, at most "
                          f"the box's per-GPU share of 16; os.cpu_count() = {os.cpu_count()})",
                "id_mismatches_vs_gpu": int((got != want["tri"]).sum()),
            }
            if spectral is not None and spectral.get("value") is not None:
                # spectral leg beside it: sample 0 of every pixel (1 of the 64 spp: ~2 M paths, ~8 M segments)
                spp_total = a.spectral_spp * world
                t1 = time.perf_counter()
                cpu_img, ccnt = o.render(osc, a.width, a.height, 0, 1, spp_total, 1, mode=abi.PT_RENDER_SUM, count=True)
                cpu_s = time.perf_counter() - t1
                ctx.render_dev(scene, a.width, a.height, 0, 1, spp_total, 1, abi.PT_RENDER_SUM, d_img.data_ptr())
                gpu_img = d_img.cpu().numpy()
                cpu_img = cpu_img.reshape(gpu_img.shape)
                same = ((gpu_img.view(np.uint32) == cpu_img.view(np.uint32)) | (np.isnan(gpu_img) & np.isnan(cpu_img))).all(axis=-1)
                spectral["cpu_baseline"] = {
                    "value": 1.0 / cpu_s, "unit": "spp/s", "cores": o.nthreads, "kind": "port", "host_cpus_visible": os.cpu_count(),
                    "sample": f"sample 0 of every pixel of the {a.width}x{a.height} frame (1 spp of {spp_total}), {cpu_s:.1f} s, "
                              f"reference-algorithm CPU restatement, {o.nthreads} threads",
                    "pixels_differing_vs_gpu": int((~same).sum()),
                }
                # algorithmic bytes per path-sample = sum over segments (B_ray + 2*128 state + 64 shading record
                # + 48 material) + 16 (pixel accumulate); visits under the reference's rule counted by the oracle on that
                # sample (every sample of a pixel is statistically the same work)
                paths, segs, nodes, tris_t = ccnt
                per_path = (segs * (32 + 16 + 2 * 128 + 64 + 48) + 64 * nodes + 48 * tris_t) / paths + 16
                total = per_path * a.width * a.height * spp_total
                gbps = total / spectral["seconds"] / 1e9
                # counter bytes of the integrator's kernels (tools/pmc_integrator.sh on this scene and frame), per path-sample
                it, itp = {}, os.path.join(ROOT, "profiles", "integrator_traffic.json")
                if os.path.exists(itp):
                    it = json.load(open(itp))
                    if not (it.get("tris") == a.tris and it.get("width") == a.width and it.get("height") == a.height and it.get("traversal") == a.traversal):
                        it = {}
                traffic = it["bytes_per_path_sample"] * a.width * a.height * spp_total if it else None
                # What binds the leg is what binds its dominant kernel, the QUEUED instance of the traversal kernel
                # (trace_share_of_device_time_rank0 of the render): its VALU issue rate and its vector-L1 lane-load rate against the
                # same two measured ceilings as the headline, counts per ray from the PMC passes of tools/pmc_integrator.sh over
                # this scene and frame, rate = the live traversal rate of this render.  SURVEY 8(d)'s HBM figures ride beside it
                # under their own names (hbm_algorithmic_frac > 1: re-reads served by the caches; it does not bind).
                qt = it.get("queued_traversal") or {}
                rays_per_s = spectral["trace_Mrays_per_s_rank0"] * 1e6
                sceil = {}
                ipath, gpath = os.path.join(ROOT, "profiles", "issue_roof.json"), os.path.join(ROOT, "profiles", "gather_roof.json")
                if qt.get("valu_instructions_per_ray") and os.path.exists(ipath):
                    ij = json.load(open(ipath))
                    rate = qt["valu_instructions_per_ray"] * rays_per_s
                    sceil["valu-issue"] = {"bound": "valu-issue", "achieved": rate, "peak": ij["wave_instr_per_s_kernel_mix"], "unit": "wave64-instr/s",
                                           "frac": rate / ij["wave_instr_per_s_kernel_mix"]}
                if qt.get("l1_accesses_per_ray") and os.path.exists(gpath):
                    gj = json.load(open(gpath))
                    rate = qt["l1_accesses_per_ray"] * rays_per_s / 1e9
                    sceil["l1-lane-loads"] = {"bound": "l1-lane-loads", "achieved": rate, "peak": gj["l1_lane_loads_per_ns"], "unit": "lane-loads/ns",
                                              "frac": rate / gj["l1_lane_loads_per_ns"]}
                if sceil:
                    top = max(sceil.values(), key=lambda c: c["frac"])
                    sroof = dict(top)
                    sroof["other_ceilings"] = {k: v for k, v in sceil.items() if v is not top}
                else:       # no matching counter profile: nothing on-chip to quote; the HBM figures below are labelled for what they are
                    sroof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None}
                sroof.update({
                    "kernel": qt.get("kernel", KERNELS[a.traversal] + " (queued)"), "kernel_share_of_device_time": spectral["trace_share_of_device_time_rank0"],
                    "valu_busy": qt.get("valu_busy"), "active_lanes_per_valu_instruction": qt.get("active_lanes_per_valu_instruction"),
                    "wave_cycles_waiting_frac": qt.get("wave_cycles_waiting_frac"), "l1_miss_lines_per_ray": qt.get("l1_miss_lines_per_ray"),
                    "valu_instructions_per_ray": qt.get("valu_instructions_per_ray"), "l1_accesses_per_ray": qt.get("l1_accesses_per_ray"),
                })
                spectral["roofline"] = sroof
                spectral["roofline"].update({
                    "hbm_algorithmic_GBps": gbps, "hbm_algorithmic_frac": gbps / 8000.0, "traffic": traffic,
                    "hbm_counter_frac": (traffic / spectral["seconds"] / 8e12) if traffic else None,
                    "traffic_bytes_per_path_sample": it.get("bytes_per_path_sample"), "traffic_by_kernel": it.get("kernels"),
                    "traffic_source": it.get("source"),
                    "algorithmic_bytes_per_path_sample": per_path, "segments_per_path": segs / paths,
                    "n_node_per_segment": nodes / segs, "n_tri_per_segment": tris_t / segs,
                    "note": "bound / achieved / peak / frac: the on-chip ceiling the leg's dominant kernel (the queued traversal kernel, "
                            "kernel_share_of_device_time of the render) sits closest to; valu_busy = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * cycles) of "
                            "that kernel, counters only. hbm_algorithmic_* is SURVEY 8(d)'s figure (reference-rule visits counted by the oracle "
                            "on sample 0 of every pixel, as bytes over wall time): > 1 of HBM peak because L1 / L2 / Infinity Cache serve the "
                            "re-reads - it does not bind. traffic = counter bytes of all the integrator's kernels for this render (per "
                            "path-sample x paths; read side doubled, Infinity-Cache hits included), hbm_counter_frac = that over wall time and 8 TB/s",
                })
        except Exception as e:  # oracle not built: keep the bench usable, say so
            cpu_baseline = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"unavailable: {type(e).__name__}: {e}"}
    if n_node is None and os.path.exists(golden):
        g = json.load(open(golden))
        if g.get("tris") == a.tris and g.get("width") == a.width and g.get("height") == a.height:
            n_node, n_tri, counts_src = g["n_node"], g["n_tri"], "tests/golden/soup1m_primary_counts.json"

    roofline = None
    if n_node is not None and not strong:
        sec = kernel_ms_avg * 1e-3
        b_ray = 32 + 16 + 64 * n_node + 48 * n_tri
        alg_gbps = b_ray * n / sec / 1e9
        # counts per launch from the PMC passes of tools/profile_bench.sh on this same command (profiles/traffic.json;
        # used only when that file matches this workload and traversal)
        tj = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if not (tj.get("traversal") == a.traversal and tj.get("tris") == a.tris and tj.get("rays") == n):
                tj = {}
        traffic = tj.get("hbm_bytes_per_launch")
        ceilings = {}
        ipath, gpath = os.path.join(ROOT, "profiles", "issue_roof.json"), os.path.join(ROOT, "profiles", "gather_roof.json")
        if tj.get("SQ_INSTS_VALU") and os.path.exists(ipath):
            ij = json.load(open(ipath))
            rate = tj["SQ_INSTS_VALU"] / sec
            ceilings["valu-issue"] = {
                "bound": "valu-issue", "achieved": rate, "peak": ij["wave_instr_per_s_kernel_mix"], "unit": "wave64-instr/s",
                "frac": rate / ij["wave_instr_per_s_kernel_mix"], "valu_wave_instructions_per_launch": tj["SQ_INSTS_VALU"],
                "active_lanes_per_valu_instruction": (tj["SQ_THREAD_CYCLES_VALU"] / tj["SQ_INSTS_VALU"]) if tj.get("SQ_THREAD_CYCLES_VALU") else None,
                "wave_cycles_waiting_frac": (tj["SQ_WAIT_ANY"] / tj["SQ_WAVE_CYCLES"]) if tj.get("SQ_WAVE_CYCLES") else None,
                # straight from the counters of the profiled launches, no measured ceiling involved: SQ_ACTIVE_INST_VALU counts
                # quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1024 SIMDs
                "valu_busy": (tj["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * tj["GRBM_GUI_ACTIVE"] / 8.0)) if tj.get("SQ_ACTIVE_INST_VALU") and tj.get("GRBM_GUI_ACTIVE") else None,
                "note": "peak = what all 1024 SIMDs issue of this kernel's instruction mix (v_min / v_max / v_cmp / v_cndmask / v_pk_*), "
                        "measured by harness/ptissue (" + ij.get("source", "profiles/") + ")"}
        if tj.get("TCP_TOTAL_CACHE_ACCESSES_sum") and os.path.exists(gpath):
            # TCP_TOTAL_CACHE_ACCESSES counts one access per active lane of a dwordx4 load (calibrated on harness/ptgather);
            # the ceiling is what that program reaches when every lane walks its own chain of L1-resident 128-byte records
            gj = json.load(open(gpath))
            rate = tj["TCP_TOTAL_CACHE_ACCESSES_sum"] / sec / 1e9
            ceilings["l1-lane-loads"] = {
                "bound": "l1-lane-loads", "achieved": rate, "peak": gj["l1_lane_loads_per_ns"], "unit": "lane-loads/ns",
                "frac": rate / gj["l1_lane_loads_per_ns"], "accesses_per_launch": tj["TCP_TOTAL_CACHE_ACCESSES_sum"],
                "l1_miss_lines_per_ns": (tj["TCP_TCC_READ_REQ_sum"] / sec / 1e9) if tj.get("TCP_TCC_READ_REQ_sum") else None,
                "random_line_ceiling_per_ns": gj.get("random_lines_per_ns"),
                "note": "ceilings measured on this GPU model by harness/ptgather (" + gj.get("source", "profiles/") + ")"}
        hbm = {"hbm_algorithmic_frac": alg_gbps / 8000.0, "hbm_algorithmic_GBps": alg_gbps,
               "hbm_counter_frac": (traffic / sec / 8e12) if traffic else None, "traffic": traffic, "traffic_source": tj.get("source"),
               "bytes_per_ray": b_ray, "n_node": n_node, "n_tri": n_tri, "visit_counts": counts_src,
               "algorithmic_bytes_per_launch": b_ray * n, "kernel": KERNELS[a.traversal], "kernel_ms": kernel_ms_avg}
        if ceilings:
            top = max(ceilings.values(), key=lambda c: c["frac"])          # the ceiling the kernel sits closest to binds
            roofline = dict(top)
            roofline["other_ceilings"] = {k: v for k, v in ceilings.items() if v is not top}
            roofline.update(hbm)
            roofline["hbm_note"] = ("hbm_algorithmic_frac is SURVEY 8(d)'s figure (every visit of the reference's traversal counted as an HBM "
                                    "read) over kernel time: > 1 because the ~100 MB node + triangle set is re-read from L1 / L2 / Infinity "
                                    "Cache; hbm_counter_frac is what the memory side really moved. HBM does not bind this kernel")
        else:
            roofline = {"bound": "hbm", "achieved": alg_gbps, "peak": 8000.0, "unit": "GB/s", "frac": alg_gbps / 8000.0}
            roofline.update(hbm)
            roofline["hbm_note"] = "no matching PMC profile under profiles/ for this workload: only SURVEY 8(d)'s algorithmic figure is shown (it does not bind)"

    if pipelined and pipelined.get("value") and roofline and roofline.get("bound") in ("valu-issue", "l1-lane-loads") and world == 1:
        # the same launches (same instruction and access counts each) in less time per launch: where the pipelined rate sits
        # against the same two measured ceilings
        gain = kernel_ms_avg / pipelined["ms_per_step"]
        pipelined["frac_of_ceilings"] = {roofline["bound"]: roofline["frac"] * gain,
                                         **{k: v["frac"] * gain for k, v in roofline["other_ceilings"].items()}}
    if rank == 0:
        out = {
            "metric": "closest-hit Mrays/s", "value": mrays, "unit": "Mrays/s", "n_gpus": world,
            "ranks_seen": ranks_seen, "backend": (dist.get_backend() if use_dist else None),
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "soup-1M primary-ray closest hit, 1920x1080 pixel centres (BASELINE configs[1])"
                if (a.tris, a.width, a.height) == (1_000_000, 1920, 1080) else
                f"soup-{a.tris} primary-ray closest hit, {a.width}x{a.height}",
                "triangles": a.tris, "rays_per_step_per_gpu": n, "traversal": a.traversal,
                "bvh_depth": info.max_depth, "scene_device_bytes": info.device_bytes,
                "host_scene_build_s": build_s,
                "sharding": ("replicated scene, the one batch cut into contiguous chunks per rank, no collective" if strong else
                             "replicated scene, one full batch per rank, no collective"),
            },
            "roofline": roofline, "cpu_baseline": cpu_baseline, "pipelined": pipelined, "other_traversal": alt, "spectral": spectral, "config4": config4,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
