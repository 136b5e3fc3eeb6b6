#!/bin/bash
# round 6: a LONG run of the randomised campaigns on the final build (new seeds), one gpurun call
# (-> gpurun_out/fuzz_r06b/*.txt; the tails go to profiles/r06b_fuzz_summary.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fuzz_r06b
mkdir -p $OUT; cd $REPO
python3 tests/fuzz/fuzz_parity.py 4000 307 both > $OUT/parity.txt 2>&1; echo "parity 4000/307: $(tail -1 $OUT/parity.txt)"
python3 tests/fuzz/fuzz_multistep.py 12000 311 > $OUT/multistep.txt 2>&1; echo "multistep 12000/311: $(tail -1 $OUT/multistep.txt)"
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python3 tests/fuzz/fuzz_multistep.py 1500 313 > $OUT/multistep_generic.txt 2>&1; echo "multistep generic 1500/313: $(tail -1 $OUT/multistep_generic.txt)"
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python3 tests/fuzz/fuzz_parity.py 600 317 both > $OUT/parity_generic.txt 2>&1; echo "parity generic 600/317: $(tail -1 $OUT/parity_generic.txt)"
python3 tests/fuzz/fuzz_kdk.py 3000 331 > $OUT/kdk.txt 2>&1; echo "kdk 3000/331: $(tail -1 $OUT/kdk.txt)"
python3 tests/fuzz/fuzz_pyexp.py 1500 337 > $OUT/pyexp.txt 2>&1; echo "pyexp 1500/337: $(tail -1 $OUT/pyexp.txt)"
python3 tests/fuzz/fuzz_store.py 4000 347 > $OUT/store.txt 2>&1; echo "store 4000/347: $(tail -1 $OUT/store.txt)"
python3 tests/fuzz/fuzz_covariance.py 1200 349 > $OUT/covariance.txt 2>&1; echo "covariance 1200/349: $(tail -1 $OUT/covariance.txt)"
python3 tests/fuzz/fuzz_orient.py 1500 353 > $OUT/orient.txt 2>&1; echo "orient 1500/353: $(tail -1 $OUT/orient.txt)"
# keep what comes back small: the mismatching lines (if any) and the summaries
for f in $OUT/*.txt; do grep -v " ok$" $f | tail -200 > $f.short; mv $f.short $f; done
