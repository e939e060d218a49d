# GPU box: config 3 (64 spp at 1080p) with the render cut into passes of different sizes and 1..4 of them in flight
set -o pipefail
mkdir -p gpurun_out
for cfg in "134217728 2" "67108864 2" "33554432 2" "33554432 4" "16777216 4" "67108864 1"; do
  set -- $cfg
  echo "== PTC_PASS_PATHS=$1 PTC_PASS_SLOTS=$2"
  PTC_PASS_PATHS=$1 PTC_PASS_SLOTS=$2 timeout -k 10 200 python tools/quick_render_bench.py --spp 64 --reps 3 2>/dev/null | grep -E "spp in|image crc" | tail -3 || exit 1
done
