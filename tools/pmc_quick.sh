#!/bin/bash
# GPU-box script: SQ counters of the traversal kernel under tools/quick_trace_bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmcq_${1:-x}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -o q -- python3 $R/tools/quick_trace_bench.py --reps 3 ${@:2} > $OUT/$name.log 2>&1 || { echo "pass failed: $pass"; tail -3 $OUT/$name.log; }
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'trace' in r['Kernel_Name']: acc[r['Kernel_Name'][26:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,cs in acc.items():
        for c,v in cs.items(): print(f"{k:46s} {c:34s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
