#!/bin/bash
# bench lines of the default library for several argument sets; usage: tools/ab_args.sh "<common args>" "<args 1>" "<args 2>" ...
common=$1; shift
mkdir -p gpurun_out/ab
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $common $a > gpurun_out/ab/args_$i.json 2>/dev/null || { echo "$a failed"; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab/args_$i.json')); r=d['roofline']
print('[$a]', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', r['kernel_ms'], 'nodes/prims per ray', r['nodes_per_ray'], r['prims_per_ray'], 'shadow', r['shadow']['nodes_per_ray'], r['shadow']['prims_per_ray'], d['bvh'])"
done
