#!/bin/bash
# GPU-box script: counters of the integrator's kernels on config 3's scene, 1 + 16 spp in two renders (17 spp of paths).
# One rocprofv3 --pmc run per counter group.  usage: tools/pmc_integrator.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; OUT=$R/gpurun_out/pmci_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
PASSES=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"
 "GRBM_GUI_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
)
for pass in "${PASSES[@]}"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -o q -- python3 $R/tools/quick_render_bench.py --spp 16 --reps 1 > $OUT/$name.log 2>&1 || { echo "pass failed: $pass"; tail -3 $OUT/$name.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $OUT k_ > $OUT/summary.txt
grep -h "spp in\|profile_totals" $OUT/FETCH_SIZE.log >> $OUT/summary.txt
python3 $R/tools/integrator_traffic.py $OUT $TAG > $OUT/integrator_traffic.json
cat $OUT/summary.txt
