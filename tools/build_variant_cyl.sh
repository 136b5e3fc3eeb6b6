#!/bin/bash
# Quick A/B variant that differs only in cyl.hip: tools/build_variant_cyl.sh <suffix> "<extra flags>"
# (reuses every other object of the main build) -> exp_amd/libexp_amd_<suffix>.so
set -e
cd "$(dirname "$0")/.."
SUF=$1; EXTRA=$2
mkdir -p build/obj_v
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics $EXTRA -c exp_amd/csrc/cyl.hip -o build/obj_v/cyl_$SUF.o 2>/dev/null
OBJS=$(ls build/obj/*.o | grep -v "/cyl.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o exp_amd/libexp_amd_$SUF.so $OBJS build/obj_v/cyl_$SUF.o -ldl
ls -la exp_amd/libexp_amd_$SUF.so
