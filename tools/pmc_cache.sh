#!/bin/bash
# L1 (TCP) and L2 (TCC) hit counters of every kernel, one rocprofv3 pass each; usage: tools/pmc_cache.sh <outdir> [bench args]
set -u
out=$1; shift
export TMPDIR=/tmp
cd /tmp
i=0
for pmc in \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $pmc --output-format csv -d "$out/pass$i" -- python3 /root/repo/bench.py --no-cpu-baseline --no-secondary "$@" > "$out.pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out.pass$i.log"; }
done
echo done
