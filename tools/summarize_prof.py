#!/usr/bin/env python3
"""Condenses rocprofv3 csv output (kernel stats + per-dispatch PMC rows) into a small text summary and
traffic_<tag>.json (memory-side bytes per launch of the traversal kernel)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
KERNEL = os.environ.get("PT_PROF_KERNEL", "k_trace_scoped_persistent")      # the dominant kernel of the default traversal
print(f"# rocprofv3 summary {tag}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print(f"## kernel stats ({os.path.relpath(f, out)})")
    for r in csv.DictReader(open(f)):
        print("  {:>8} calls  avg {:>12.1f} ns  total {:>14} ns  {:>6}%  {}".format(
            r.get("Calls"), float(r.get("AverageNs", 0)), r.get("TotalDurationNs"), r.get("Percentage"), r.get("Name", "")[:100]))
means = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"## counters ({os.path.basename(d)})")
        for k, cs in acc.items():
            if "trace" not in k and "shade" not in k:
                continue
            for c, v in cs.items():
                print(f"  {k[:60]:60s} {c:30s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
                if KERNEL in k:
                    means[c] = sum(v) / len(v)
                    means["kernel"] = k[k.index("k_trace"):].split("(")[0]
if "FETCH_SIZE" in means:
    t = {
        "tag": tag, "kernel": means.get("kernel"),
        "FETCH_SIZE_KB": means.get("FETCH_SIZE"), "WRITE_SIZE_KB": means.get("WRITE_SIZE"),
        "TCC_EA0_RDREQ": means.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_32B": means.get("TCC_EA0_RDREQ_32B_sum"),
        "TCC_EA0_RDREQ_64B": means.get("TCC_EA0_RDREQ_64B_sum"), "TCC_EA0_RDREQ_128B": means.get("TCC_EA0_RDREQ_128B_sum"),
        "TCC_EA0_RDREQ_DRAM": means.get("TCC_EA0_RDREQ_DRAM_sum"), "TCC_EA0_WRREQ": means.get("TCC_EA0_WRREQ_sum"),
        "TCC_EA0_WRREQ_64B": means.get("TCC_EA0_WRREQ_64B_sum"),
        "TCC_HIT": means.get("TCC_HIT_sum"), "TCC_MISS": means.get("TCC_MISS_sum"),
    }
    # issue side (SQ counters; *_CYCLES and ACTIVE_INST_* are in quad-cycles per SIMD-wave slot, summed over the chip)
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_THREAD_CYCLES_VALU",
              "SQ_WAVES", "GRBM_GUI_ACTIVE", "TA_BUSY_avr", "TCP_TCP_LATENCY_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"):
        if c in means:
            t[c] = means[c]
    # MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B and reads exactly half of the
    # bytes of a wide (16 B/lane) read because 128-B requests are tallied at 64 B -> double the read side;
    # WRITE_SIZE is exact.  Infinity-Cache hits are counted, so this is an UPPER bound of HBM bytes.
    if t["FETCH_SIZE_KB"] is not None and t["WRITE_SIZE_KB"] is not None:
        t["hbm_bytes_per_launch"] = (2.0 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024.0
        t["correction"] = "read side doubled (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); Infinity-Cache hits included"
    # what bench.py matches this file against (tools/profile_bench.sh runs bench.py's default workload)
    t["traversal"] = os.environ.get("PT_PROF_TRAVERSAL", "exact")
    t["tris"] = int(os.environ.get("PT_PROF_TRIS", "1000000"))
    t["rays"] = int(os.environ.get("PT_PROF_RAYS", "2073600"))
    t["source"] = (f"profiles/{tag}_summary.txt (rocprofv3 --pmc passes of `bench.py --steps 20 --warmup 2 --no-cpu-baseline "
                   f"--spectral-spp 0 --config4-spp 0 --in-flight 1`, one pass per counter group, mean over the launches of the traversal kernel)")
    json.dump(t, open(os.path.join(out, f"traffic_{tag}.json"), "w"), indent=1)
    print("## traffic", json.dumps(t))
