#!/bin/bash
# SQ / TA counters of every kernel, two rocprofv3 passes; usage: tools/pmc_sq.sh <outdir> [bench args]
set -u
out=$1; shift
export TMPDIR=/tmp
cd /tmp
i=0
for pmc in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $pmc --output-format csv -d "$out/pass$i" -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > "$out.pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out.pass$i.log"; }
done
echo done
