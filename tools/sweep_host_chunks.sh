# GPU box: pt_trace_closest (host pointers, pinned) against chunk size and the number of trace slots its kernels alternate between
set -o pipefail
for cfg in "1048576 1" "1048576 2" "524288 2" "262144 2" "262144 3" "131072 3" "524288 1"; do
  set -- $cfg
  echo "== PTC_HOST_CHUNK=$1 PTC_HOST_SLOTS=$2"
  PTC_HOST_CHUNK=$1 PTC_HOST_SLOTS=$2 timeout -k 10 200 python -u tools/quick_trace_bench.py --fast-only 2>/dev/null | grep -E "host-pointer" | cut -c1-250 || exit 1
done
