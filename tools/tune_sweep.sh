#!/bin/bash
# usage: tools/tune_sweep.sh "<bench args>" tune1 tune2 ...   (prints Mrays/s and kernel ms per tune value)
args=$1; shift
for t in "$@"; do
  out=$(timeout -k 10 200 python bench.py $args --no-cpu-baseline --tune $t 2>/dev/null | tail -1)
  echo "tune=$t $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])')"
done
