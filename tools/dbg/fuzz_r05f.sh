#!/bin/bash
# the campaigns that exercise the sort passes of a level range, on the round's last build (after the short-range variants of
# k_key_hist / k_scatter_adv): one gpurun call, new seeds (-> gpurun_out/fuzz_r05f/*.txt -> profiles/r05f_fuzz_*.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fuzz_r05f
mkdir -p $OUT; cd $REPO
python3 tests/fuzz/fuzz_multistep.py 1500 331 > $OUT/multistep.txt 2>&1; tail -1 $OUT/multistep.txt
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python3 tests/fuzz/fuzz_multistep.py 400 337 > $OUT/multistep_generic.txt 2>&1; tail -1 $OUT/multistep_generic.txt
EXP_AMD_SIM_OVERLAP=0 python3 tests/fuzz/fuzz_multistep.py 400 347 > $OUT/multistep_one_stream.txt 2>&1; tail -1 $OUT/multistep_one_stream.txt
python3 tests/fuzz/fuzz_kdk.py 300 349 > $OUT/kdk.txt 2>&1; tail -1 $OUT/kdk.txt
python3 tests/fuzz/fuzz_store.py 300 353 > $OUT/store.txt 2>&1; tail -1 $OUT/store.txt
python3 tools/dbg/soak_cfg4.py 2e6 200 > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
