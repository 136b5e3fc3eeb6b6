#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh <suffix> "<extra hipcc flags>"
# -> exp_amd/libexp_amd_<suffix>.so (select it with EXP_AMD_LIB=...)
set -e
cd "$(dirname "$0")/.."
SUF=$1; EXTRA=$2
OBJ=build/obj_$SUF
mkdir -p $OBJ
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics $EXTRA"
pids=()
for f in context particles sph sph_fields cyl host force_api; do
  hipcc $FLAGS -c exp_amd/csrc/$f.hip -o $OBJ/$f.o 2>/dev/null &
  pids+=($!)
done
for l in 0 1 2 3 4 5 6 7 8 9 10 11 12; do
  hipcc $FLAGS -DSPH_L=$l -c exp_amd/csrc/sph_inst.hip -o $OBJ/sph_inst_L$l.o 2>/dev/null &
  pids+=($!)
  if (( ${#pids[@]} >= 8 )); then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o exp_amd/libexp_amd_$SUF.so $OBJ/*.o -ldl
ls -la exp_amd/libexp_amd_$SUF.so
