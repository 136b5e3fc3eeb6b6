#!/bin/bash
# round 6, final build: the randomised campaigns once more, new seeds, the fused-step and store campaigns also with the append
# step forced on small components (full and lean payload, with and without slack) -- one gpurun call
# (-> gpurun_out/fuzz_r06c/*.txt; the tails go to profiles/r06c_fuzz_summary.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fuzz_r06${TAG:-c}
O=${SEED_OFF:-0}     # (added to every seed: a second pass of the same campaigns with other draws)
mkdir -p $OUT; cd $REPO
python3 tests/fuzz/fuzz_parity.py 1500 $((401+O)) both > $OUT/parity.txt 2>&1; echo "parity 1500/401: $(tail -1 $OUT/parity.txt)"
python3 tests/fuzz/fuzz_multistep.py 6000 $((409+O)) > $OUT/multistep.txt 2>&1; echo "multistep 6000/409: $(tail -1 $OUT/multistep.txt)"
python3 tests/fuzz/fuzz_kdk.py 3000 $((419+O)) > $OUT/kdk.txt 2>&1; echo "kdk 3000/419: $(tail -1 $OUT/kdk.txt)"
EXP_AMD_APPEND_MIN=200 python3 tests/fuzz/fuzz_kdk.py 2000 $((421+O)) > $OUT/kdk_append.txt 2>&1; echo "kdk append 2000/421: $(tail -1 $OUT/kdk_append.txt)"
EXP_AMD_APPEND_MIN=200 EXP_AMD_APPEND_LEAN=1 python3 tests/fuzz/fuzz_kdk.py 2000 $((431+O)) > $OUT/kdk_lean.txt 2>&1; echo "kdk append lean 2000/431: $(tail -1 $OUT/kdk_lean.txt)"
EXP_AMD_APPEND_MIN=-200 EXP_AMD_APPEND_LEAN=1 python3 tests/fuzz/fuzz_kdk.py 1000 $((433+O)) > $OUT/kdk_lean_tight.txt 2>&1; echo "kdk append lean no-slack 1000/433: $(tail -1 $OUT/kdk_lean_tight.txt)"
EXP_AMD_APPEND_MIN=-200 python3 tests/fuzz/fuzz_kdk.py 1000 $((439+O)) > $OUT/kdk_tight.txt 2>&1; echo "kdk append no-slack 1000/439: $(tail -1 $OUT/kdk_tight.txt)"
EXP_AMD_APPEND_MIN=200 python3 tests/fuzz/fuzz_store.py 3000 $((443+O)) > $OUT/store_append.txt 2>&1; echo "store append 3000/443: $(tail -1 $OUT/store_append.txt)"
python3 tests/fuzz/fuzz_pyexp.py 600 $((449+O)) > $OUT/pyexp.txt 2>&1; echo "pyexp 600/449: $(tail -1 $OUT/pyexp.txt)"
python3 tests/fuzz/fuzz_covariance.py 600 $((457+O)) > $OUT/covariance.txt 2>&1; echo "covariance 600/457: $(tail -1 $OUT/covariance.txt)"
python3 tests/fuzz/fuzz_orient.py 500 $((461+O)) > $OUT/orient.txt 2>&1; echo "orient 500/461: $(tail -1 $OUT/orient.txt)"
for f in $OUT/*.txt; do grep -v " ok$" $f | tail -200 > $f.short; mv $f.short $f; done
