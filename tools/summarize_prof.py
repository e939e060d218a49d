#!/usr/bin/env python3
"""Condenses rocprofv3 csv output (kernel stats + per-dispatch PMC rows) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
print(f"# rocprofv3 summary {tag}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print(f"## kernel stats ({os.path.relpath(f, out)})")
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        print("  {:>8} calls  avg {:>12.1f} ns  total {:>14} ns  {:>6}%  {}".format(
            r.get("Calls"), float(r.get("AverageNs", 0)), r.get("TotalDurationNs"), r.get("Percentage"), r.get("Name", "")[:90]))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"## counters ({os.path.basename(d)})")
        for k, cs in acc.items():
            if "trace" not in k and "shade" not in k:
                continue
            for c, v in cs.items():
                print(f"  {k:60s} {c:28s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
