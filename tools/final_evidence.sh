#!/bin/bash
# Everything the round's profiles are made from, for the tree as it is: GPU tests, bench lines of all workloads,
# kernel-trace stats, counter passes (HBM traffic, SQ, TA) for the headline workload in both semantics.
# usage: tools/final_evidence.sh <tag>     (run from the repository root on the GPU box; outputs under gpurun_out/<tag>/)
set -e
tag=$1
o=gpurun_out/$tag
mkdir -p $o
./tools/round_check.sh $tag
for sem in 1 0; do
  ./tools/pmc_passes.sh $PWD/$o/pmc_dragon_s$sem --semantics $sem
  python3 tools/pmc_summary.py $o/pmc_dragon_s$sem dragon $sem $o/pmc_counters.json $o/dragon_s${sem}_pmc_counters.md $tag
  rm -rf $o/pmc_dragon_s$sem/pass?
done
cp $o/pmc_counters.json pmc_counters.json
timeout -k 10 300 python bench.py 2>/dev/null > $o/bench_dragon.json
cut -c1-200 $o/bench_dragon.json
timeout -k 10 300 python bench.py --semantics 0 2>/dev/null > $o/bench_dragon_literal.json
cut -c1-200 $o/bench_dragon_literal.json
for w in cornell cornell_ggx sphere dragon_4k; do
  timeout -k 10 400 python bench.py --workload $w 2>/dev/null > $o/bench_$w.json
  cut -c1-200 $o/bench_$w.json
done
