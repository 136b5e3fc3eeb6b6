#!/bin/bash
# A/B variant that differs in ONE translation unit: tools/build_variant_tu.sh <suffix> <tu> "<extra hipcc flags>"
#   <tu> = a file of exp_amd/csrc without .hip (particles, sph, cyl, host ...), or sph_inst_L<k> for one harmonic order
# Reuses every other object of the main build (make lib first) -> exp_amd/libexp_amd_<suffix>.so (select with EXP_AMD_LIB)
set -e
cd "$(dirname "$0")/.."
SUF=$1; TU=$2; EXTRA=$3
mkdir -p build/obj_v
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics $EXTRA"
if [[ $TU == sph_inst_L* ]]; then
  hipcc $FLAGS -DSPH_L=${TU#sph_inst_L} -c exp_amd/csrc/sph_inst.hip -o build/obj_v/${TU}_$SUF.o 2>/dev/null
else
  hipcc $FLAGS -c exp_amd/csrc/$TU.hip -o build/obj_v/${TU}_$SUF.o 2>/dev/null
fi
OBJS=$(ls build/obj/*.o | grep -v "/$TU.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o exp_amd/libexp_amd_$SUF.so $OBJS build/obj_v/${TU}_$SUF.o -ldl
ls -la exp_amd/libexp_amd_$SUF.so
