#!/bin/bash
# second set of the round's campaigns (final build; new seeds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/fuzz
run() { out=gpurun_out/fuzz/$1; shift; "$@" > $out 2>&1; tail -1 $out; }
run r04b_fuzz_multistep.txt python tests/fuzz/fuzz_multistep.py 2000 83
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 run r04b_fuzz_multistep_generic.txt python tests/fuzz/fuzz_multistep.py 600 89
EXP_AMD_THIN_V=2 run r04b_fuzz_multistep_thinv2.txt python tests/fuzz/fuzz_multistep.py 600 97
EXP_AMD_SIM_EARLY_CROSS=1 EXP_AMD_ACC_SIDE=1 run r04b_fuzz_multistep_optional.txt python tests/fuzz/fuzz_multistep.py 600 101
run r04b_fuzz_parity.txt python tests/fuzz/fuzz_parity.py 400 83 both
run r04b_fuzz_kdk.txt python tests/fuzz/fuzz_kdk.py 300 83
run r04b_fuzz_store.txt python tests/fuzz/fuzz_store.py 300 83
run r04b_fuzz_pyexp.txt python tests/fuzz/fuzz_pyexp.py 200 83
