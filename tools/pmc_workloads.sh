#!/bin/bash
# Counter passes (tools/pmc_passes.sh) for further workloads, merged into pmc_counters.json so that their bench lines
# carry the VALU-issue and HBM fractions too.  usage: tools/pmc_workloads.sh <tag> <workload> ...   (GPU box, repository root)
set -e
tag=$1; shift
o=gpurun_out/$tag
mkdir -p $o
for w in "$@"; do
  ./tools/pmc_passes.sh $PWD/$o/pmc_$w --workload $w
  python3 tools/pmc_summary.py $o/pmc_$w $w 1 pmc_counters.json $o/${w}_pmc_counters.md $tag > /dev/null
  rm -rf $o/pmc_$w/pass?
  timeout -k 10 400 python bench.py --workload $w 2>/dev/null > $o/bench_$w.json
  cut -c1-160 $o/bench_$w.json
done
cp pmc_counters.json $o/pmc_counters.json
