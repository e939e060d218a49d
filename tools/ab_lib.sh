# GPU box: A/B of alternative builds of the library (PTCORE_LIB) on config 2 primaries, bounce rays and config 3.
# usage: tools/ab_lib.sh libptcore.so libptcore_other.so ...
set -o pipefail
mkdir -p gpurun_out
for lib in "${@:-libptcore.so}"; do
  echo "== $lib"
  export PTCORE_LIB=$PWD/pathtracer_amd/$lib
  timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only 2>/dev/null | grep "exact:" || exit 1
  timeout -k 10 120 python -u tools/quick_trace_bench.py --device-only --fast-only --bounce 2>/dev/null | grep "exact:" || exit 1
  timeout -k 10 200 python tools/quick_render_bench.py --spp 64 --reps 2 | tail -2 || exit 1
done
