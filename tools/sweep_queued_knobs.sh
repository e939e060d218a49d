for cfg in "PTC_X=0" "PTC_SAMPLE_GROUP=32" "PTC_SAMPLE_GROUP=64" "PTC_SAMPLE_GROUP=8" "PTC_TICKET=128" "PTC_TICKET=32" "PTC_BLOCKS_PER_CU=4" "PTC_SAMPLE_GROUP=32 PTC_TICKET=128" "PTC_REFILL_QUEUED=28" "PTC_REFILL_QUEUED=36" "PTC_X=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/quick_render_bench.py --spp 64 --reps 3 2>/dev/null | grep -E "spp in" | tail -1 | cut -c1-110 || exit 1
done
