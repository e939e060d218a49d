#!/bin/bash
# GPU-box script: hardware counters of the traversal kernel under tools/quick_trace_bench.py,
# one rocprofv3 --pmc run per pass (rocprofv3 does not split counters over passes).
# usage: tools/pmc_quick.sh <tag> [quick_trace_bench args...]; PTC_* env selects the kernel variant
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmcq_${1:-x}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum"
 "TA_BUSY_avr TA_TA_BUSY_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum"
 "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_REQ_sum TCC_HIT_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_READ_sum"
 "GRBM_GUI_ACTIVE GRBM_TA_BUSY"
)
for pass in "${PASSES[@]}"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -o q -- python3 $R/tools/quick_trace_bench.py --reps 3 --fast-only ${@:2} > $OUT/$name.log 2>&1 || { echo "pass failed: $pass"; tail -3 $OUT/$name.log; }
done
python3 - <<PY > $OUT/summary.txt
import csv,glob,collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'trace' in r['Kernel_Name']: acc[r['Kernel_Name'][26:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,cs in acc.items():
        for c,v in cs.items(): print(f"{k:46s} {c:38s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
cat $OUT/summary.txt
