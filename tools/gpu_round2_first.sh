#!/bin/bash
# first GPU contact of round 2: the rewritten traversal kernels (parity tests), the allocation-free entry point, timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_trace.py tests/test_golden.py tests/wip_no_alloc.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/first_tests.log 2>&1
rc=$?
tail -15 gpurun_out/first_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/quick_trace_bench.py > gpurun_out/first_bench.log 2>&1 || { tail -20 gpurun_out/first_bench.log; exit 1; }
cat gpurun_out/first_bench.log
timeout -k 10 300 python -u tools/quick_trace_bench.py --bounce > gpurun_out/first_bench_bounce.log 2>&1 || { tail -20 gpurun_out/first_bench_bounce.log; exit 1; }
cat gpurun_out/first_bench_bounce.log
