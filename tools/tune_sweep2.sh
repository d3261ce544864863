#!/bin/bash
# Sweep on the dragon workload; each argument is  <tune>[@<variant>]  (variant = a library
# built by tools/build_variant.sh); one bench line per argument.
out=gpurun_out/tune_sweep2.txt
: > $out
for a in "$@"; do
  t=${a%@*}; v=""; [[ "$a" == *@* ]] && v=${a#*@}
  echo "tune $a" >> $out
  if [ -n "$v" ]; then export PTCORE_LIB=$PWD/pathtracer_amd/_lib/variants/libptcore_$v.so; else unset PTCORE_LIB; fi
  timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --tune $t 2>/dev/null >> $out || exit 1
done
