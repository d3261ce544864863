#!/bin/bash
# Bench lines of the default library and of variants built by tools/build_variant.sh, same box, back to back.
# usage: tools/ab.sh "<bench args>" base|<variant> ...     (outputs under gpurun_out/ab/)
args=$1; shift
mkdir -p gpurun_out/ab
for tag in "$@"; do
  if [ $tag = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$PWD/pathtracer_amd/_lib/variants/libptcore_$tag.so; fi
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $args > gpurun_out/ab/$tag.json 2>gpurun_out/ab/$tag.err || { echo "$tag failed"; tail -3 gpurun_out/ab/$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab/$tag.json')); r=d['roofline']
print('$tag', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', r['kernel_ms'], 'batch', d['batch'])"
done
