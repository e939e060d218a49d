#!/bin/bash
# GPU-box script: hardware counters of kernels whose name contains <filter>, for any python tool.
# usage: tools/pmc_any.sh <tag> <filter> <script.py> [args...]   (one rocprofv3 --pmc run per pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; FILTER=$2; shift 2
OUT=$R/gpurun_out/pmca_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE GRBM_TA_BUSY"
)
for pass in "${PASSES[@]}"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -o q -- python3 $R/$@ > $OUT/$name.log 2>&1 || { echo "pass failed: $pass"; tail -3 $OUT/$name.log; }
done
python3 - <<PY > $OUT/summary.txt
import csv,glob,collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "$FILTER" in r['Kernel_Name']: acc[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,cs in acc.items():
        for c,v in cs.items(): print(f"{k:42s} {c:34s} n={len(v):3d} sum={sum(v):.5g} mean={sum(v)/len(v):.5g}")
PY
cat $OUT/summary.txt
