"""The JSON line bench.py prints (driver contract): checked on the committed output of the last GPU run
(profiles/r01i_bench.json is what `python bench.py` printed on the MI355X box) and on bench.py's own source."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_key():
    j = json.load(open(os.path.join(ROOT, "profiles", "r01i_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"] == "closest-hit Mrays/s" and j["unit"] == "Mrays/s" and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] < r["algorithmic_bytes_per_launch"]
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["id_mismatches_vs_gpu"] == 0
    assert abs(j["value"] - j["config"]["rays_per_step_per_gpu"] * j["n_gpus"] / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * j["value"]
    s = j["spectral"]
    assert s["unit"] == "spp/s" and s["cpu_baseline"]["pixels_differing_vs_gpu"] == 0


def test_bench_reads_nothing_from_the_reference_at_run_time():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "/root/reference" not in src
    assert re.search(r"--gpus", src) and re.search(r"--steps", src) and re.search(r"--warmup", src)
