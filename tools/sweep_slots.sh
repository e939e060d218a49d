#!/bin/bash
# GPU box: config 3 / config 4 shapes under different pass schedules (PTC_PASS_SLOTS x PTC_PASS_PATHS x PTC_ADMIT_PERCENT)
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/sweep_slots.txt; : > $out
run() { echo "## $*" | tee -a $out; env "${@:2}" timeout -k 10 300 python tools/quick_render_bench.py $1 --reps 2 2>&1 | tail -2 | tee -a $out || exit 1; }
M=1048576
for shape in "--spp 256" "--width 3840 --height 2160 --spp 64"; do
run "$shape" PTC_PASS_SLOTS=1 PTC_PASS_PATHS=$((128*M))
run "$shape" PTC_PASS_SLOTS=2 PTC_PASS_PATHS=$((64*M)) PTC_ADMIT_PERCENT=10
run "$shape" PTC_PASS_SLOTS=2 PTC_PASS_PATHS=$((64*M)) PTC_ADMIT_PERCENT=25
run "$shape" PTC_PASS_SLOTS=3 PTC_PASS_PATHS=$((43*M)) PTC_ADMIT_PERCENT=25
run "$shape" PTC_PASS_SLOTS=4 PTC_PASS_PATHS=$((32*M)) PTC_ADMIT_PERCENT=40
run "$shape" PTC_PASS_SLOTS=2 PTC_PASS_PATHS=$((96*M)) PTC_ADMIT_PERCENT=10
run "$shape" PTC_PASS_SLOTS=3 PTC_PASS_PATHS=$((64*M)) PTC_ADMIT_PERCENT=25
run "$shape" PTC_PASS_SLOTS=4 PTC_PASS_PATHS=$((48*M)) PTC_ADMIT_PERCENT=40
done
