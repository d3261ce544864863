#!/bin/bash
# Hardware counters of one bench.py command, one rocprofv3 --pmc pass per counter group (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; 8 SQ counters per pass).  The bench line of every pass goes to its log,
# which tools/pmc_summary.py reads for the ray counts.
# usage: tools/pmc_passes.sh <outdir> [bench args]   (run from the repository root)
set -u
out=$1; shift
root=$PWD
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
i=0
for pmc in \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
  "SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/pass$i" -- python3 $root/bench.py --no-cpu-baseline --no-secondary --profile-min-paths 1 --steps 1 --warmup 0 "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
  echo "pass $i ok"
done
