#!/bin/bash
# GPU-box script: A/B the traversal kernel knobs on soup-1M 1080p rays.
# usage: [EXTRA="--bounce"] tools/sweep_trace.sh "variant blocks_per_cu refill ring" ...
for cfg in "$@"; do
  set -- $cfg
  echo "variant=$1 blocks_per_cu=$2 refill=$3 ring=$4"
  PTC_TRACE_VARIANT=$1 PTC_BLOCKS_PER_CU=$2 PTC_REFILL=$3 PTC_RING=$4 timeout -k 10 120 python -u tools/quick_trace_bench.py --reps 10 --fast-only $EXTRA 2>&1 | grep -E "^fast|^wide" || exit 1
done
