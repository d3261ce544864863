#!/bin/bash
# Kernel timeline of the last timed bench step: start offset, duration and queue of every kernel longer than 0.2 ms.
# usage: tools/diag/timeline.sh [bench args]     (output: gpurun_out/timeline.txt)
root=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tl && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $root/gpurun_out/timeline.log 2>&1 || { tail -5 $root/gpurun_out/timeline.log; exit 1; }
python3 - $root/gpurun_out/timeline.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:44], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
gen = [i for i, r in enumerate(rows) if r[2].startswith('k_generate')]
rows = rows[gen[-1]:]   # the last batch
t0 = rows[0][0]
with open(sys.argv[1], 'w') as o:
    last_end = {}
    for s, e, name, q, st in rows:
        if (e - s) < 200000 and not name.startswith('k_'): continue
        line = '%9.3f ms  +%8.3f ms  q%s s%s  %s' % ((s - t0) / 1e6, (e - s) / 1e6, q, st, name)
        print(line); o.write(line + '\n')
PY
