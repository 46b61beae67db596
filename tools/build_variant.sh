#!/bin/bash
# usage: tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>
# Builds perception_amd/lib/variants/lib<name>.so = the current objects with <source.hip> recompiled with the extra flags
# (cost probes, experiments).  Run `make -C perception_amd/csrc` first.
set -e
cd "$(dirname "$0")/../perception_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p build/var ../lib/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o build/var/$name.o
objs=""
for o in build/*.o; do
  if [ "$(basename $o .o)" = "$(basename $src .hip)" ]; then objs="$objs build/var/$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/lib$name.so $objs
echo ../lib/variants/lib$name.so
