# GPU box: A/B of alternative builds of the library (PTCORE_LIB) on config 2 primaries and bounce rays (and, with RENDER=1,
# the integrator).  usage: tools/ab_lib.sh libptcore.so libptcore_other.so ...  (PTC_* tuning variables pass through)
set -o pipefail
mkdir -p gpurun_out
for lib in "${@:-libptcore.so}"; do
  echo "== $lib"
  export PTCORE_LIB=$PWD/pathtracer_amd/$lib
  timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only --reps 20 2>/dev/null | grep -E "exact|equal" || exit 1
  timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only --reps 20 --bounce 2>/dev/null | grep -E "exact|equal" || exit 1
  if [ -n "$RENDER" ]; then timeout -k 10 200 python tools/quick_render_bench.py --spp 64 --reps 2 | tail -2 || exit 1; fi
done
