#!/bin/bash
# GPU tests, full bench, kernel-trace stats of a short bench; outputs under gpurun_out/<tag>*.
# usage: tools/round_check.sh <tag>
set -e
tag=$1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${tag}_gpu_tests.log 2>&1
tail -2 gpurun_out/${tag}_gpu_tests.log
timeout -k 10 300 python bench.py 2>/dev/null > gpurun_out/${tag}_bench.json
cat gpurun_out/${tag}_bench.json
rm -rf gpurun_out/${tag}_prof
export TMPDIR=/tmp
root=$PWD
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_prof -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_prof.log 2>&1
echo profiled
