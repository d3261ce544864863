#!/bin/bash
# Build libptcore with extra compile flags into pathtracer_amd/_lib/variants/libptcore_<name>.so
# usage: tools/build_variant.sh <name> [-DKNOB=value ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=pathtracer_amd/_lib/variants; tmp=build/variant_$name
mkdir -p $out $tmp
for f in wavefront bvh_build bvh_device ray_sort ptcore; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack- -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Iinclude -Ipathtracer_amd/csrc -Wno-unused-function "$@" \
    -c pathtracer_amd/csrc/$f.hip -o $tmp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950:xnack- -shared -fPIC -o $out/libptcore_$name.so $tmp/wavefront.o $tmp/bvh_build.o $tmp/bvh_device.o $tmp/ray_sort.o $tmp/ptcore.o
echo built $out/libptcore_$name.so
