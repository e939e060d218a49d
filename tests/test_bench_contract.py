"""The JSON line bench.py prints (driver contract): checked on the committed output of the last GPU run
(profiles/r03e_bench.json is what `python bench.py` printed on the MI355X box on the round's final tree; the counter files it is
checked against - profiles/traffic.json, integrator_traffic.json, r03e_summary.txt - are the PMC passes of that same tree; the
earlier r03a - r03d lines stay under profiles/ as records of their own trees) and on bench.py's own source."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["r03e_bench.json"])
def test_committed_bench_line_has_every_contract_key(name):
    j = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "backend", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["ranks_seen"] == 1 and j["backend"] is None
    assert j["metric"] == "closest-hit Mrays/s" and j["unit"] == "Mrays/s" and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["traversal"] == "exact"
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_algorithmic_frac", "hbm_counter_frac", "other_ceilings"):
        assert k in r, k
    # the primary fraction is the ceiling that binds (VALU issue or the vector L1's lane-load rate), an efficiency below 1;
    # SURVEY 8(d)'s HBM figures ride beside it, labelled for what they are
    assert r["bound"] in ("valu-issue", "l1-lane-loads") and 0.3 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    for o in r["other_ceilings"].values():
        assert o["frac"] <= r["frac"] and abs(o["frac"] - o["achieved"] / o["peak"]) < 1e-9
    assert abs(r["hbm_algorithmic_frac"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 8e12) < 1e-6
    assert r["traffic"] < 0.05 * r["algorithmic_bytes_per_launch"] and abs(r["hbm_counter_frac"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 8e12) < 1e-9
    # every count behind the fractions is reproducible from the committed profile of the same command
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["traversal"] == "exact" and t["rays"] == j["config"]["rays_per_step_per_gpu"] and t["kernel"].startswith("k_trace_scoped_persistent")
    assert r["traffic"] == t["hbm_bytes_per_launch"]
    # valu_busy comes straight from the counters: SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
    vi = r if r["bound"] == "valu-issue" else r["other_ceilings"]["valu-issue"]
    assert abs(vi["valu_busy"] - t["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * t["GRBM_GUI_ACTIVE"] / 8)) < 1e-9 and 0.4 < vi["valu_busy"] < 0.9
    ij = json.load(open(os.path.join(ROOT, "profiles", "issue_roof.json")))
    assert vi["peak"] == ij["wave_instr_per_s_kernel_mix"] and "tools/issue_mix.py" in ij["source"] and set(ij["hot_block_mix"]) >= {"record_sign", "leaf", "pop"}
    summary = open(os.path.join(ROOT, "profiles", "r03e_summary.txt")).read()
    assert "k_trace_scoped_persistent" in summary and "SQ_INSTS_VALU" in summary and "TCP_TOTAL_CACHE_ACCESSES_sum" in summary
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "host_cpus_visible"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["id_mismatches_vs_gpu"] == 0 and 1 <= c["cores"] <= c["host_cpus_visible"]
    assert abs(j["value"] - j["config"]["rays_per_step_per_gpu"] * j["n_gpus"] / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * j["value"]
    assert j["other_traversal"]["traversal"] == "min_t" and j["other_traversal"]["ids_differing_from_exact"] == 0
    if name >= "r03c":
        # the same K batches with two in flight (pt_trace_closest_dev_submit): beside the headline, never instead of it; same hits;
        # its fractions of the two ceilings are the headline's scaled by the time per launch, and stay below 1
        pl = j["pipelined"]
        assert pl["unit"] == "Mrays/s" and pl["in_flight"] == 2 and pl["steps"] == j["steps"] and pl["hits_equal_blocking_call"] is True
        assert abs(pl["value"] - j["config"]["rays_per_step_per_gpu"] / (pl["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * pl["value"]
        assert j["value"] < pl["value"] < 1.6 * j["value"]
        fc = pl["frac_of_ceilings"]
        assert set(fc) == {"valu-issue", "l1-lane-loads"} and all(0.3 < v < 1.0 for v in fc.values())
        assert abs(fc[r["bound"]] - r["frac"] * r["kernel_ms"] / pl["ms_per_step"]) < 1e-9
        s3 = open(os.path.join(ROOT, "profiles", "r03e_summary.txt")).read()
        assert "k_trace_scoped_persistent<12, false>" in s3 and "23 calls" in s3      # every launch in the trace is one launch alone (--in-flight 1)
    s = j["spectral"]
    assert s["unit"] == "spp/s" and s["cpu_baseline"]["pixels_differing_vs_gpu"] == 0 and s["scaling"] == "weak"
    sr = s["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_path_sample", "kernel", "kernel_share_of_device_time",
              "valu_busy", "hbm_algorithmic_frac", "hbm_algorithmic_GBps", "hbm_counter_frac", "other_ceilings"):
        assert k in sr, k
    # what is printed under `bound` binds: an on-chip ceiling of the leg's dominant kernel (the QUEUED traversal instance), a
    # fraction below 1; SURVEY 8(d)'s HBM figure rides beside it under its own name and may exceed 1
    assert sr["bound"] in ("valu-issue", "l1-lane-loads") and 0.3 < sr["frac"] < 1.0 and abs(sr["frac"] - sr["achieved"] / sr["peak"]) < 1e-9
    assert all(o["frac"] <= sr["frac"] < 1.0 for o in sr["other_ceilings"].values()) and sr["kernel"] == "k_trace_scoped_persistent<12, true>"
    assert 0.6 < sr["kernel_share_of_device_time"] < 0.95 and 0.3 < sr["valu_busy"] < 0.9
    assert abs(sr["hbm_algorithmic_GBps"] - sr["algorithmic_bytes_per_path_sample"] * 1920 * 1080 * s["spp_total"] / s["seconds"] / 1e9) < 1e-6 * sr["hbm_algorithmic_GBps"]
    d = s["device_ms_by_kernel_rank0"]
    assert 0.9 * d["total"] < d["trace"] + d["shade"] + d["sky"] + d["accumulate"] + d["generate"] <= d["total"]

    def no_frac_above_one_under_a_bound(o):
        if isinstance(o, dict):
            if "bound" in o and o.get("frac") is not None:
                assert o["frac"] <= 1.0, o
            for v in o.values():
                no_frac_above_one_under_a_bound(v)
    no_frac_above_one_under_a_bound(j)
    # the counter bytes of the integrator's kernels come from the committed PMC passes (tools/pmc_integrator.sh)
    it = json.load(open(os.path.join(ROOT, "profiles", "integrator_traffic.json")))
    assert sr["traffic_bytes_per_path_sample"] == it["bytes_per_path_sample"] and sr["traffic"] < sr["algorithmic_bytes_per_path_sample"] * 1920 * 1080 * s["spp_total"]
    assert abs(sr["hbm_counter_frac"] - sr["traffic"] / s["seconds"] / 8e12) < 1e-9 and set(it["kernels"]) >= {"k_shade", "k_trace_scoped_persistent", "k_sky", "k_accumulate"}
    assert it["queued_traversal"]["kernel"] == sr["kernel"] and sr["valu_busy"] == it["queued_traversal"]["valu_busy"]
    assert os.path.exists(os.path.join(ROOT, it["source"].split(" ")[0]))
    ns = s["normalized_sky_direction"]          # the opt-in extension beside the faithful leg: no NaN pixel, about the same rate
    assert ns["finite_pixel_fraction"] == 1.0 and s["finite_pixel_fraction"] < 0.1 and 0.8 * s["value"] < ns["value"] < 1.25 * s["value"]
    c4 = j["config4"]
    assert c4["unit"] == "spp/s" and c4["scaling"] == "strong" and c4["spp_total"] == 1024 and c4["allreduce_bytes"] == 3840 * 2160 * 16
    assert abs(c4["value"] - c4["spp_total"] / c4["seconds"]) < 1e-9 * c4["value"]


def test_bench_reads_nothing_from_the_reference_at_run_time():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "/root/reference" not in src
    assert re.search(r"--gpus", src) and re.search(r"--steps", src) and re.search(r"--warmup", src)


def test_bench_starts_its_own_ranks_and_counts_them():
    """`python bench.py --gpus 2` - the plain command, no torchrun - starts two ranks as a child process before anything touches
    the GPU; --launch-check stops after the rendezvous (gloo here: no GPU), and the line says how many ranks the group holds."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j == {"launch_check": True, "n_gpus": 2, "world_size": 2, "ranks_seen": 2, "backend": "gloo"}
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr


def test_launcher_command_is_a_child_torchrun_and_never_an_exec():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    cmd = b.launcher_command(["--gpus", "8", "--steps", "3"], 8, 12345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src
    # the launch happens before torch is imported in the launcher process
    main_src = src[src.index("def main():"):]
    assert main_src.index("self_launch(a") < main_src.index("import torch")
    # the line carries the rank count and keeps the CPU baseline at N > 1
    assert '"ranks_seen": ranks_seen' in src and "if rank == 0 and not a.no_cpu_baseline" in src
