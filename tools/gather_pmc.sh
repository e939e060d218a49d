#!/bin/bash
# GPU box: calibrates the L1 counters against harness/ptgather, whose lane-load count is known.
# Usage: tools/gather_pmc.sh <records>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gather_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 60 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcp -o g -- $R/harness/ptgather --records ${1:-64} --steps 64 > $OUT/tcp.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/tcp.log; exit 1; }
timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d $OUT/sq -o g -- $R/harness/ptgather --records ${1:-64} --steps 64 > $OUT/sq.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/sq.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for sub in ("tcp", "sq"):
    for f in glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        by = collections.OrderedDict()
        for r in rows:
            by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:24]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for (d, k), v in by.items():
            if (d - 1) % 4 == 3:      # last repetition of each configuration
                print(sub, d, k, v)
PY
