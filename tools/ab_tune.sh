#!/bin/bash
# bench lines of one library for several PTC_OPT_TUNE_TRAVERSAL values; usage: tools/ab_tune.sh base|<variant> "<bench args>" <tune> ...
tag=$1; shift
args=$1; shift
if [ $tag = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$PWD/pathtracer_amd/_lib/variants/libptcore_$tag.so; fi
mkdir -p gpurun_out/ab
for t in "$@"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $args --tune $t > gpurun_out/ab/${tag}_$t.json 2>/dev/null || { echo "$tag $t failed"; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab/${tag}_$t.json')); r=d['roofline']
print('$tag', '$t', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', r['kernel_ms'])"
done
