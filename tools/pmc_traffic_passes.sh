#!/bin/bash
# FETCH_SIZE and WRITE_SIZE of the default bench workload, one rocprofv3 pass each.
set -u
out=$1; shift
export TMPDIR=/tmp
cd /tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/pass$i" -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > "$out.pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out.pass$i.log"; exit 1; }
done
echo done
