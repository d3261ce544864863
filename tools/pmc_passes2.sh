#!/bin/bash
# Memory-pipeline counters (TA / TCP), one rocprofv3 pass per hardware block budget.
set -u
out=$1; shift
export TMPDIR=/tmp
cd /tmp
i=0
for pmc in \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $pmc --output-format csv -d "$out/pass$i" -- python3 /root/repo/bench.py --no-cpu-baseline --no-secondary "$@" > "$out.pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out.pass$i.log"; }
done
echo done
