#!/bin/bash
# GPU box: the whole -m gpu suite, then (optionally) the named extra test files alone in a fresh process each
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests "$@" -x -q -m gpu --timeout 600 --timeout-method thread > gpurun_out/suite.log 2>&1
rc=$?
tail -12 gpurun_out/suite.log
[ $rc -ne 0 ] && exit $rc
for f in "$@"; do
  timeout -k 10 300 python -u -m pytest "$f" -x -q -m gpu --timeout 200 --timeout-method thread > gpurun_out/alone_$(basename $f .py).log 2>&1 || { tail -20 gpurun_out/alone_$(basename $f .py).log; exit 1; }
  tail -2 gpurun_out/alone_$(basename $f .py).log
done
