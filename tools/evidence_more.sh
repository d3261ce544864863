#!/bin/bash
# Second half of a round's evidence (after tools/final_evidence.sh <tag>): counter passes of the other workloads,
# kernel timelines of one dragon frame and of one Cornell GGX batch, the hierarchy build alone, how many lanes of a
# wave share a node.  usage: tools/evidence_more.sh <tag>   (GPU box, repository root; outputs under gpurun_out/<tag>/)
set -e
tag=$1
o=gpurun_out/$tag
mkdir -p $o
# (the top-level pmc_counters.json holds the first half's entries: gpurun_out/ does not travel to the box)
./tools/pmc_workloads.sh $tag cornell cornell_ggx sphere dragon_4k
./tools/diag/timeline.sh > /dev/null && cp gpurun_out/timeline.txt $o/dragon_timeline.txt
./tools/diag/timeline.sh --workload cornell_ggx > /dev/null && cp gpurun_out/timeline.txt $o/cornell_ggx_timeline.txt
(for b in 3 4 1 2; do python3 tools/diag/build_only.py $b; done) > $o/build_times.txt 2>&1
(python3 tools/diag/wave_coherence.py 1; python3 tools/diag/wave_coherence.py 5) > $o/wave_coherence.txt 2>&1
# the default run once more, now that every workload has its counters: the line the profiles quote
timeout -k 10 400 python bench.py 2>/dev/null > $o/bench_dragon.json
cut -c1-200 $o/bench_dragon.json
