#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc passes tools/pmc_any.sh leaves under gpurun_out/pmca_<tag>/.
usage: tools/pmc_summary.py <dir> [name-filter]"""
import collections
import csv
import glob
import re
import sys

d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        if not m or flt not in m.group(0):
            continue
        acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:34s} n={len(v):4d} mean={sum(v)/len(v):.6g} sum={sum(v):.6g}")
