#!/bin/bash
# bench lines with hierarchy statistics for the default library and variants; usage: tools/ab_tree.sh base|<variant> ...
for tag in "$@"; do
  if [ $tag = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$PWD/pathtracer_amd/_lib/variants/libptcore_$tag.so; fi
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ab_$tag.json 2>/dev/null || { echo "$tag failed"; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); r=d['roofline']
print('$tag', d['value'], d['ms_per_step'], r['kernel_ms'], 'ext nodes/prims', r['nodes_per_ray'], r['prims_per_ray'], 'shadow', r['shadow']['nodes_per_ray'], r['shadow']['prims_per_ray'], d['bvh'])"
done
