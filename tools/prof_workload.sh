#!/bin/bash
# rocprofv3 kernel-trace stats of one bench workload; usage: tools/prof_workload.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p $root/gpurun_out/prof
cd /tmp
rm -rf $root/gpurun_out/prof/$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof/$tag -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $root/gpurun_out/prof/$tag.log 2>&1 || { echo failed; tail -5 $root/gpurun_out/prof/$tag.log; exit 1; }
cd $root
python3 tools/summarize_prof.py gpurun_out/prof/$tag gpurun_out/prof/$tag.md "$tag: bench.py --steps 2 --warmup 1 $*"
rm -rf gpurun_out/prof/$tag
grep '"metric"' gpurun_out/prof/$tag.log | cut -c1-160
head -16 gpurun_out/prof/$tag.md
