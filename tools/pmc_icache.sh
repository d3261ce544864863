#!/bin/bash
# Instruction-cache counters of one bench.py step, summed per kernel.  usage: tools/pmc_icache.sh <outdir> [bench args]
set -u
out=$(realpath -m "$1"); shift
root=$PWD
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$out/avail.txt" 2>&1
grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_WAIT_INST_ANY" "$out/avail.txt" | sort -u | tr '\n' ' ' > "$out/names.txt"
echo "available: $(cat $out/names.txt)"
timeout -s KILL 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$out/pass" -- python3 $root/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 "$@" > "$out/pass.log" 2>&1 || { echo "pass failed"; tail -5 "$out/pass.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/pass/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
with open(out + '/icache.md', 'w') as o:
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:12]:
        line = k + ': ' + ', '.join('%s=%.4g' % (n, v) for n, v in sorted(c.items()))
        print(line); o.write(line + '\n')
PY
rm -rf "$out/pass"
