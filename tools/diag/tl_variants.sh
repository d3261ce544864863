#!/bin/bash
# first lines of the frame timeline for the default library and variants; usage: tools/diag/tl_variants.sh base|<variant> ...
for v in "$@"; do
  if [ $v = base ]; then unset PTCORE_LIB; else export PTCORE_LIB=$PWD/pathtracer_amd/_lib/variants/libptcore_$v.so; fi
  tools/diag/timeline.sh > /dev/null || exit 1
  echo "== $v"; head -5 gpurun_out/timeline.txt
done
