for r in 16 24 32; do
  echo "== PTC_REFILL=$r"
  PTC_REFILL=$r timeout -k 10 400 python -u tools/run_big_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('  ', j['leg'][:40], {k: round(v, 1) for k, v in j.items() if k in ('spp_per_s', 'trace_Mrays_per_s', 'Mrays_per_s', 'bounce_Mrays_per_s')})
" || exit 1
done
