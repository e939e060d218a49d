for cfg in "PTC_REFILL=16" "PTC_REFILL=4" "PTC_REFILL=8" "PTC_REFILL=12" "PTC_REFILL=24" "PTC_REFILL=32" "PTC_TICKET=32" "PTC_TICKET=128" "PTC_RING=10" "PTC_RING=8"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only 2>/dev/null | grep "exact:" || exit 1
  env $cfg timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only --bounce 2>/dev/null | grep "exact:" || exit 1
done
