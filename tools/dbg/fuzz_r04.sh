#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/fuzz
python tests/fuzz/fuzz_multistep.py 1500 41 > gpurun_out/fuzz/r04_fuzz_multistep.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_multistep.txt
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python tests/fuzz/fuzz_multistep.py 600 43 > gpurun_out/fuzz/r04_fuzz_multistep_generic.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_multistep_generic.txt
EXP_AMD_THIN_V=2 python tests/fuzz/fuzz_multistep.py 600 47 > gpurun_out/fuzz/r04_fuzz_multistep_thinv2.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_multistep_thinv2.txt
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python tests/fuzz/fuzz_parity.py 400 41 both > gpurun_out/fuzz/r04_fuzz_parity_generic.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_parity_generic.txt
python tests/fuzz/fuzz_parity.py 400 43 both > gpurun_out/fuzz/r04_fuzz_parity.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_parity.txt
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python tests/fuzz/fuzz_kdk.py 300 41 > gpurun_out/fuzz/r04_fuzz_kdk_generic.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_kdk_generic.txt
python tests/fuzz/fuzz_kdk.py 300 43 > gpurun_out/fuzz/r04_fuzz_kdk.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_kdk.txt
python tests/fuzz/fuzz_covariance.py 300 41 > gpurun_out/fuzz/r04_fuzz_covariance.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_covariance.txt
python tests/fuzz/fuzz_pyexp.py 300 41 > gpurun_out/fuzz/r04_fuzz_pyexp.txt 2>&1; tail -1 gpurun_out/fuzz/r04_fuzz_pyexp.txt
