for wh in "3840 2160" "7680 4320"; do
  set -- $wh
  for r in 16 24 32; do
    echo "== ${1}x${2} PTC_REFILL=$r"
    PTC_REFILL=$r timeout -k 10 200 python -u tools/quick_trace_bench.py --device-only --fast-only --reps 8 --width $1 --height $2 2>/dev/null | grep "exact:" || exit 1
  done
done
