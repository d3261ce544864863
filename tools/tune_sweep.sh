#!/bin/bash
# A/B sweep of bench.py; each argument is  <tune>[@<variant>]  where <tune> is a
# PTC_OPT_TUNE_TRAVERSAL value and <variant> a library built by tools/build_variant.sh.
# One bench line per argument in gpurun_out/tune_sweep.txt.  WORKLOAD=dragon|cornell|... (default dragon).
out=gpurun_out/tune_sweep.txt
: > $out
for a in "$@"; do
  t=${a%@*}; v=""; [[ "$a" == *@* ]] && v=${a#*@}
  echo "tune $a" >> $out
  if [ -n "$v" ]; then export PTCORE_LIB=$PWD/pathtracer_amd/_lib/variants/libptcore_$v.so; else unset PTCORE_LIB; fi
  timeout -k 10 200 python bench.py --workload ${WORKLOAD:-dragon} --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --tune $t 2>/dev/null >> $out || exit 1
done
