#!/bin/bash
# round 5's randomised campaigns on the final build, one gpurun call (-> gpurun_out/fuzz_r05/*.txt -> profiles/r05_fuzz_*.txt):
# the multistep campaign now draws the option keys (rtrunc / com0, ton / toff / twid, self_consistent, FIX_L0, mlim) on half
# of its trials
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fuzz_r05
mkdir -p $OUT; cd $REPO
python3 tests/fuzz/fuzz_multistep.py 1500 ${SEED0:-97} > $OUT/multistep.txt 2>&1; tail -1 $OUT/multistep.txt
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python3 tests/fuzz/fuzz_multistep.py 400 101 > $OUT/multistep_generic.txt 2>&1; tail -1 $OUT/multistep_generic.txt
EXP_AMD_SIM_OVERLAP=0 python3 tests/fuzz/fuzz_multistep.py 400 103 > $OUT/multistep_one_stream.txt 2>&1; tail -1 $OUT/multistep_one_stream.txt
python3 tests/fuzz/fuzz_parity.py 300 107 both > $OUT/parity.txt 2>&1; tail -1 $OUT/parity.txt
python3 tests/fuzz/fuzz_kdk.py 300 109 > $OUT/kdk.txt 2>&1; tail -1 $OUT/kdk.txt
python3 tests/fuzz/fuzz_pyexp.py 150 113 > $OUT/pyexp.txt 2>&1; tail -1 $OUT/pyexp.txt
python3 tests/fuzz/fuzz_store.py 300 127 > $OUT/store.txt 2>&1; tail -1 $OUT/store.txt
