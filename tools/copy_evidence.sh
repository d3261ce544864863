#!/bin/bash
# Copy what tools/final_evidence.sh + tools/pmc_workloads.sh + the two timelines left under gpurun_out/<tag>/ into profiles/<tag>_*
# and pmc_counters.json.  usage: tools/copy_evidence.sh <tag>
set -e
tag=$1; o=gpurun_out/$tag
cp $o/pmc_counters.json pmc_counters.json
cp $o/bench_dragon.json profiles/${tag}_bench_dragon.json
cp $o/bench_dragon_literal.json profiles/${tag}_bench_dragon_literal.json
for w in cornell cornell_ggx sphere dragon_4k; do
  cp $o/bench_$w.json profiles/${tag}_bench_$w.json
  cp $o/${w}_pmc_counters.md profiles/${tag}_${w}_pmc_counters.md
done
cp $o/kernel_stats.md profiles/${tag}_dragon_kernel_stats.md
cp $o/dragon_s1_pmc_counters.md profiles/${tag}_dragon_pmc_counters.md
cp $o/dragon_s0_pmc_counters.md profiles/${tag}_dragon_literal_pmc_counters.md
(echo "# $tag: kernel timeline of one frame (bench.py --steps 1 --warmup 1, dragon 1080p x 256 spp): start offset, duration, queue, stream"; echo; echo '```'; cat $o/dragon_timeline.txt; echo '```') > profiles/${tag}_dragon_timeline.md
(echo "# $tag: kernel timeline of the last batch of one frame (bench.py --steps 1 --warmup 1 --workload cornell_ggx): start offset, duration, queue, stream; kernels shorter than 0.2 ms left out"; echo; echo '```'; cat $o/cornell_ggx_timeline.txt; echo '```') > profiles/${tag}_cornell_ggx_timeline.md
for w in dragon dragon_literal cornell cornell_ggx sphere dragon_4k; do python3 - $tag $w <<'PY'
import json,sys
d=json.loads(open('profiles/%s_bench_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'lanes', r.get('active_lane_frac'), 'valu/ray', r.get('valu_insts_per_ray'), 'hbm_frac', r.get('hbm_frac'), 'launch ms', r['avg_launch_ms'], 'total', d['total_mrays_per_s'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
done
