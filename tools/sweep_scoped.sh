#!/bin/bash
# GPU box: tuning knobs of the default (EXACT) traversal kernel on soup-1M primaries and bounce rays
mkdir -p gpurun_out
for cfg in "PTC_RING=12" "PTC_RING=8" "PTC_RING=10" "PTC_RING=14" "PTC_RING=16" "PTC_RING=12 PTC_BLOCKS_PER_CU=4" "PTC_RING=12 PTC_BLOCKS_PER_CU=5" "PTC_RING=8 PTC_BLOCKS_PER_CU=8" "PTC_RING=12 PTC_REFILL=8" "PTC_RING=12 PTC_REFILL=24" "PTC_RING=12 PTC_REFILL=32"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only 2>/dev/null | grep "exact:" || exit 1
  env $cfg timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only --bounce 2>/dev/null | grep "exact:" || exit 1
done
