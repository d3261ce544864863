#!/bin/bash
# usage: tools/tune_sweep3.sh <workload> <tune>...   one bench line per tune value
w=$1; shift
out=gpurun_out/tune_sweep3.txt
: > $out
for t in "$@"; do
  echo "tune $t" >> $out
  timeout -k 10 200 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --tune $t 2>/dev/null >> $out || exit 1
done
