for cfg in "PTC_REFILL_X=0" "PTC_REFILL=12 PTC_REFILL_QUEUED=32" "PTC_REFILL=20 PTC_REFILL_QUEUED=32" "PTC_REFILL=24 PTC_REFILL_QUEUED=32" "PTC_REFILL=16 PTC_REFILL_QUEUED=24" "PTC_REFILL=16 PTC_REFILL_QUEUED=28" "PTC_REFILL_X=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/quick_render_bench.py --spp 64 --reps 3 2>/dev/null | grep -E "spp in" | tail -1 | cut -c1-60 || exit 1
  env $cfg timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only --reps 20 2>/dev/null | grep "exact:" || exit 1
  env $cfg timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only --reps 20 --bounce 2>/dev/null | grep "exact:" || exit 1
done
