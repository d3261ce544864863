#!/bin/bash
# Launch-by-launch durations of one kernel for the default library and one variant, same box.
# usage: tools/prof_launches_ab.sh <variant> <kernel substring> [bench args]
v=$1; k=$2; shift; shift
export TMPDIR=/tmp
root=$PWD
cd /tmp
for tag in base $v; do
  if [ $tag = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$root/pathtracer_amd/_lib/variants/libptcore_$tag.so; fi
  rm -rf $root/gpurun_out/pl_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/pl_$tag -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $root/gpurun_out/pl_$tag.log 2>&1 || exit 1
  echo "== $tag"; python3 $root/tools/diag/surface_launches.py $root/gpurun_out/pl_$tag "$k" 8
  rm -rf $root/gpurun_out/pl_$tag
done
