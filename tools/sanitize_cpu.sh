#!/bin/bash
# CPU sanitizer pass (AddressSanitizer + UndefinedBehaviorSanitizer) over the C code that runs on the host: the oracle
# (oracle/*.c) and the HDF5 readers / writers of the product (exp_amd/csrc_host/*.c), driven by the CPU half of the test
# suite.  (GPU sanitizers are not available on this pool.)  Rebuilds both libraries instrumented with ROCm's clang -- gcc
# 11's libasan cannot place its shadow memory under this kernel's address-space randomisation --, runs the tests with the
# runtime preloaded, then rebuilds the libraries as they were.    tools/sanitize_cpu.sh [pytest args]
set -u
cd "$(dirname "$0")/.."
CL=/opt/rocm/lib/llvm/bin/clang
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SAN="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan"
HDF5_INC=$(grep -m1 "^HDF5_INC" Makefile | sed 's/.*= *//')
HDF5_LIB=$(grep -m1 "^HDF5_LIB" Makefile | sed 's/.*= *//')
restore() {
  make -s -C oracle clean; make -s -C oracle > /dev/null
  rm -f exp_amd/libexp_amd_h5.so; make -s h5 > /dev/null
}
trap restore EXIT
make -s -C oracle clean
mkdir -p oracle/_build
# (tuned_cpu.c, the speed baseline, is instrumented as well here)
$CL $SAN -fPIC -std=gnu11 -Wall -fno-fast-math -ffp-contract=off -shared -o oracle/_build/liboracle.so oracle/bfe_oracle.c oracle/cyl_oracle.c \
    oracle/nbody_oracle.c oracle/pyexp_oracle.c oracle/refstruct_cpu.c oracle/psp_oracle.c oracle/tuned_cpu.c -lm -lpthread || exit 1
rm -f exp_amd/libexp_amd_h5.so
$CL $SAN -fPIC -shared -Wall -I$HDF5_INC exp_amd/csrc_host/h5cache.c exp_amd/csrc_host/h5part.c \
    -o exp_amd/libexp_amd_h5.so -L$HDF5_LIB -lhdf5 -Wl,-rpath,$HDF5_LIB || exit 1
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_golden_cpu.py::test_oracle_matches_sph_golden "$@"
# (deselected: that test holds the oracle to the BITS gcc -O2 produced for the committed fixture; clang -O1 differs in the
# last bit of a velocity -- a compiler difference, not a sanitizer finding)
