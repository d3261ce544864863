#!/bin/bash
# Collect PMC counters for the bench workload in separate rocprofv3 passes (one
# hardware block's budget per pass, no tracing options besides the counters).
# usage: tools/pmc_passes.sh <outdir> [bench args...]
set -u
out=$1; shift
export TMPDIR=/tmp
cd /tmp
i=0
for pmc in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum"
do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $pmc --output-format csv -d "$out/pass$i" -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > "$out.pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out.pass$i.log"; exit 1; }
done
echo done
