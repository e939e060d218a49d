#!/bin/bash
# GPU box: pop iterations per trip (PTC_POP_CAP) x lanes at a leaf before the triangle test runs (PTC_LEAF_MIN), EXACT kernel
mkdir -p gpurun_out
out=gpurun_out/sweep_lanes.txt; : > $out
for cfg in "PTC_POP_CAP=0 PTC_LEAF_MIN=1" "PTC_POP_CAP=1 PTC_LEAF_MIN=1" "PTC_POP_CAP=2 PTC_LEAF_MIN=1" "PTC_POP_CAP=3 PTC_LEAF_MIN=1" \
           "PTC_POP_CAP=0 PTC_LEAF_MIN=8" "PTC_POP_CAP=0 PTC_LEAF_MIN=16" "PTC_POP_CAP=2 PTC_LEAF_MIN=8" "PTC_POP_CAP=2 PTC_LEAF_MIN=12" "PTC_POP_CAP=2 PTC_LEAF_MIN=16" "PTC_POP_CAP=2 PTC_LEAF_MIN=24" \
           "PTC_POP_CAP=1 PTC_LEAF_MIN=16" "PTC_POP_CAP=3 PTC_LEAF_MIN=16"; do
  echo "== $cfg" | tee -a $out
  env $cfg timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only 2>/dev/null | grep "exact:" | tee -a $out || exit 1
  env $cfg timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only --bounce 2>/dev/null | grep "exact:" | tee -a $out || exit 1
done
