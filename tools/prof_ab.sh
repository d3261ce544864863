#!/bin/bash
# Kernel-trace stats of the default library and of one variant, same box; usage: tools/prof_ab.sh <variant> [bench args]
v=$1; shift
export TMPDIR=/tmp
root=$PWD
cd /tmp
for tag in base $v; do
  if [ $tag = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$root/pathtracer_amd/_lib/variants/libptcore_$tag.so; fi
  rm -rf $root/gpurun_out/ab_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/ab_$tag -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $root/gpurun_out/ab_$tag.log 2>&1 || exit 1
  python3 $root/tools/summarize_prof.py $root/gpurun_out/ab_$tag $root/gpurun_out/ab_$tag.md $tag
  rm -rf $root/gpurun_out/ab_$tag
  head -14 $root/gpurun_out/ab_$tag.md
done
