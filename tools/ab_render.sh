# GPU box: A/B of library builds on config 3 (64 spp), spp/s lines
for lib in "$@"; do
  echo "== $lib"
  PTCORE_LIB=$PWD/pathtracer_amd/$lib timeout -k 10 200 python tools/quick_render_bench.py --spp 64 --reps 3 2>/dev/null | grep -E "spp in|crc" | cut -c1-120 || exit 1
done
