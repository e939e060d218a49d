#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then one --pmc run per
# counter pass (rocprofv3 does not split counters over passes), then the summaries the repo keeps
# under profiles/ (copy them there by hand after the call: gpurun_out/ is scratch).
# Usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (--in-flight 1: no pipelined leg, so that every launch of the traversal kernel in the trace is one launch alone on the chip)
ARGS="--steps 20 --warmup 2 --no-cpu-baseline --config4-spp 0 --in-flight 1 $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $R/bench.py $ARGS --spectral-spp 16 > $OUT/trace.log 2>&1 || { echo "kernel-trace failed"; tail -5 $OUT/trace.log; exit 1; }
PASSES=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
 "GRBM_GUI_ACTIVE GRBM_TA_BUSY"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
 "TA_BUSY_avr TA_TA_BUSY_sum"
)
for pass in "${PASSES[@]}"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o $TAG -- python3 $R/bench.py $ARGS --spectral-spp 0 > $OUT/pmc_$name.log 2>&1 || { echo "pmc pass $pass failed"; tail -3 $OUT/pmc_$name.log; exit 1; }
done
python3 $R/tools/summarize_prof.py $OUT $TAG > $OUT/summary_$TAG.txt 2>&1
cat $OUT/summary_$TAG.txt
