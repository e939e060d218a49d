#!/usr/bin/env python3
"""Memory-side bytes per path-sample of the integrator's kernels, from the FETCH_SIZE / WRITE_SIZE passes that
tools/pmc_integrator.sh leaves under gpurun_out/pmci_<tag>/ (1 + 16 spp of config 3's scene at 1920x1080 = 17 spp of paths).
usage: tools/integrator_traffic.py <dir> <tag> > profiles/integrator_traffic.json"""
import collections
import csv
import glob
import json
import re
import sys

d, tag = sys.argv[1], sys.argv[2]
paths = 17 * 1920 * 1080
rays = None
for f in glob.glob(d + "/*.log"):                       # quick_render_bench.py prints what the process traced in all
    for line in open(f, errors="replace"):
        m = re.match(r"profile_totals rays (\d+) paths (\d+)", line)
        if m:
            rays, paths = int(m.group(1)), int(m.group(2))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
ISSUE = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE",
         "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum")
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if m and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") + ISSUE:
            acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
kernels = {}
for k in ("k_generate", "k_trace_scoped_persistent", "k_trace_replay_persistent", "k_shade", "k_sky", "k_accumulate", "k_output"):
    if k in acc:
        # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM): read side doubled; WRITE_SIZE exact
        kernels[k] = {"read_bytes_per_path_sample": 2.0 * acc[k]["FETCH_SIZE"] * 1024.0 / paths,
                      "write_bytes_per_path_sample": acc[k]["WRITE_SIZE"] * 1024.0 / paths}
total = sum(v["read_bytes_per_path_sample"] + v["write_bytes_per_path_sample"] for v in kernels.values())
# the QUEUED instance of the traversal kernel (k_trace_scoped_persistent<12, true>: the integrator's launches), summed over
# the whole process: what its issue side did per ray it traced
queued = None
tk = acc.get("k_trace_scoped_persistent", {})
if rays and tk.get("SQ_INSTS_VALU") and tk.get("GRBM_GUI_ACTIVE"):
    cycles = tk["GRBM_GUI_ACTIVE"] / 8.0                                 # summed over the 8 XCDs
    queued = {
        "kernel": "k_trace_scoped_persistent<12, true>", "rays": rays,
        "valu_instructions_per_ray": tk["SQ_INSTS_VALU"] / rays,
        "l1_accesses_per_ray": tk["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays if tk.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
        "l1_miss_lines_per_ray": tk["TCP_TCC_READ_REQ_sum"] / rays if tk.get("TCP_TCC_READ_REQ_sum") else None,
        "valu_busy": tk["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),   # SQ_ACTIVE_INST_VALU * 4 / (SIMDs * cycles), counters only
        "active_lanes_per_valu_instruction": tk["SQ_THREAD_CYCLES_VALU"] / tk["SQ_INSTS_VALU"] if tk.get("SQ_THREAD_CYCLES_VALU") else None,
        "wave_cycles_waiting_frac": tk["SQ_WAIT_ANY"] / tk["SQ_WAVE_CYCLES"] if tk.get("SQ_WAIT_ANY") and tk.get("SQ_WAVE_CYCLES") else None,
        "gpu_cycles": cycles,
    }
json.dump({"tag": tag, "queued_traversal": queued, "workload": "soup-1000000 + Hosek-Wilkie sky, 1920x1080, 1 + 16 spp (tools/quick_render_bench.py --spp 16 --reps 1)",
           "tris": 1000000, "width": 1920, "height": 1080, "traversal": "exact",
           "path_samples": paths, "bytes_per_path_sample": total, "kernels": kernels,
           "correction": "read side doubled (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); Infinity-Cache hits included: an upper bound of HBM bytes",
           "source": f"profiles/{tag}_pmc_integrator.txt (tools/pmc_integrator.sh {tag}: one rocprofv3 --pmc run per counter group)"},
          sys.stdout, indent=1)
