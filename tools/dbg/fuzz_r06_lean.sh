#!/bin/bash
# round 6, second session: the fused-step and store campaigns with the append step forced on from 200 / 500 particles and its LEAN
# payload (exp_amd_ctx_set_append_lean: acceleration and potential re-evaluated for the first call that looks), with and
# without slack in the regions (-> gpurun_out/fuzz_r06_lean/*.txt -> profiles/r06_fuzz_lean.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fuzz_r06_lean
mkdir -p $OUT; cd $REPO
export EXP_AMD_APPEND_LEAN=1
EXP_AMD_APPEND_MIN=200 python3 tests/fuzz/fuzz_kdk.py 400 311 > $OUT/kdk_lean.txt 2>&1; tail -1 $OUT/kdk_lean.txt
EXP_AMD_APPEND_MIN=-500 python3 tests/fuzz/fuzz_kdk.py 300 313 > $OUT/kdk_lean_tight.txt 2>&1; tail -1 $OUT/kdk_lean_tight.txt
EXP_AMD_APPEND_MIN=200 python3 tests/fuzz/fuzz_store.py 300 317 > $OUT/store_lean.txt 2>&1; tail -1 $OUT/store_lean.txt
unset EXP_AMD_APPEND_LEAN
EXP_AMD_APPEND_MIN=200 python3 tests/fuzz/fuzz_kdk.py 300 331 > $OUT/kdk_full.txt 2>&1; tail -1 $OUT/kdk_full.txt
