#!/bin/bash
# round 6's randomised campaigns on the final build, one gpurun call (-> gpurun_out/fuzz_r06/*.txt -> profiles/r06_fuzz_*.txt):
# new seeds; the parity campaign now draws the "ssfrac" key on a fifth of its sphere trials
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fuzz_r06
mkdir -p $OUT; cd $REPO
python3 tests/fuzz/fuzz_parity.py 600 211 both > $OUT/parity.txt 2>&1; tail -1 $OUT/parity.txt
python3 tests/fuzz/fuzz_multistep.py 1000 223 > $OUT/multistep.txt 2>&1; tail -1 $OUT/multistep.txt
EXP_AMD_SIM_OVERLAP=0 python3 tests/fuzz/fuzz_multistep.py 300 227 > $OUT/multistep_one_stream.txt 2>&1; tail -1 $OUT/multistep_one_stream.txt
python3 tests/fuzz/fuzz_kdk.py 300 229 > $OUT/kdk.txt 2>&1; tail -1 $OUT/kdk.txt
python3 tests/fuzz/fuzz_pyexp.py 200 233 > $OUT/pyexp.txt 2>&1; tail -1 $OUT/pyexp.txt
python3 tests/fuzz/fuzz_store.py 300 239 > $OUT/store.txt 2>&1; tail -1 $OUT/store.txt
