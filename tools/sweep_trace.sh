#!/bin/bash
# GPU-box script: A/B the traversal kernel variants / knobs on soup-1M 1080p primaries.
for cfg in "0 8 16" "1 8 16" "1 8 8" "1 8 24" "1 8 32" "1 4 16" "1 6 16" "1 2 16"; do
  set -- $cfg
  echo "variant=$1 blocks_per_cu=$2 refill=$3"
  PTC_TRACE_VARIANT=$1 PTC_BLOCKS_PER_CU=$2 PTC_REFILL=$3 timeout -k 10 120 python -u tools/quick_trace_bench.py --reps 10 $EXTRA 2>&1 | grep -E "^fast|ids" || exit 1
done
