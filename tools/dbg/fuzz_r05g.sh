#!/bin/bash
# every campaign once more on the round's last build, larger and with new seeds (one gpurun call; -> profiles/r05g_fuzz_*.txt, tails only
# for the long ones), behind __graft_entry__.smoke()
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fuzz_r05g
mkdir -p $OUT; cd $REPO
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
python3 tests/fuzz/fuzz_multistep.py 5000 359 > $OUT/multistep.txt 2>&1; tail -1 $OUT/multistep.txt
python3 tests/fuzz/fuzz_parity.py 1500 367 both > $OUT/parity.txt 2>&1; tail -1 $OUT/parity.txt
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python3 tests/fuzz/fuzz_parity.py 500 373 both > $OUT/parity_generic.txt 2>&1; tail -1 $OUT/parity_generic.txt
python3 tests/fuzz/fuzz_pyexp.py 400 379 > $OUT/pyexp.txt 2>&1; tail -1 $OUT/pyexp.txt
python3 tests/fuzz/fuzz_covariance.py 500 383 > $OUT/covariance.txt 2>&1; tail -1 $OUT/covariance.txt
python3 tests/fuzz/fuzz_orient.py 1000 389 > $OUT/orient.txt 2>&1; tail -1 $OUT/orient.txt
python3 tests/fuzz/fuzz_kdk.py 1000 397 > $OUT/kdk.txt 2>&1; tail -1 $OUT/kdk.txt
python3 tests/fuzz/fuzz_store.py 1500 401 > $OUT/store.txt 2>&1; tail -1 $OUT/store.txt
