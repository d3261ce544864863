#!/bin/bash
# Everything the round's profiles are made from, for the tree as it is: GPU tests, bench lines of all workloads,
# kernel-trace stats, HBM traffic passes, SQ/TA counter passes.  usage: tools/final_evidence.sh <tag>
set -e
tag=$1
./tools/round_check.sh $tag
mkdir -p gpurun_out/${tag}_traffic gpurun_out/${tag}_sq
./tools/pmc_traffic_passes.sh $PWD/gpurun_out/${tag}_traffic --steps 1 --warmup 0
python3 tools/pmc_traffic.py gpurun_out/${tag}_traffic dragon gpurun_out/${tag}_pmc_traffic.json
rm -rf gpurun_out/${tag}_traffic
./tools/pmc_sq.sh $PWD/gpurun_out/${tag}_sq --spp 32 --steps 1 --warmup 0
python3 tools/summarize_pmc.py gpurun_out/${tag}_sq gpurun_out/${tag}_pmc_sq.md "dragon, 32 spp"
rm -rf gpurun_out/${tag}_sq
for w in cornell cornell_ggx sphere; do
  timeout -k 10 300 python bench.py --workload $w 2>/dev/null > gpurun_out/${tag}_bench_$w.json
  cut -c1-200 gpurun_out/${tag}_bench_$w.json
done
