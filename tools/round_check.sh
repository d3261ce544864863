#!/bin/bash
# GPU tests, full bench, kernel-trace stats of a short bench; outputs under gpurun_out/<tag>/.
# usage: tools/round_check.sh <tag>
set -e
tag=$1
o=gpurun_out/$tag
mkdir -p $o
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $o/gpu_tests.log 2>&1
tail -2 $o/gpu_tests.log
timeout -k 10 300 python bench.py 2>/dev/null > $o/bench.json
cut -c1-300 $o/bench.json
rm -rf $o/prof
export TMPDIR=/tmp
root=$PWD
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/prof -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $root/$o/prof.log 2>&1
cd $root
python3 tools/summarize_prof.py $o/prof $o/kernel_stats.md "$tag: bench.py --steps 2 --warmup 1 (dragon 1080p x 256 spp), rocprofv3 --kernel-trace --stats"
rm -rf $o/prof
echo profiled
