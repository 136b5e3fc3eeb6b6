#!/bin/bash
# the round's closing call: the whole GPU suite, the multistep campaign (level-policy keys drawn) on both schedules and on the any-order
# kernels, then the profile set (prof_r05.sh) -- on the build the round ends with
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_r05
mkdir -p $OUT; cd $REPO
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1
python3 tests/fuzz/fuzz_multistep.py 3000 433 > $OUT/multistep.txt 2>&1; tail -1 $OUT/multistep.txt
EXP_AMD_SIM_OVERLAP=0 python3 tests/fuzz/fuzz_multistep.py 600 439 > $OUT/multistep_one_stream.txt 2>&1; tail -1 $OUT/multistep_one_stream.txt
EXP_AMD_SPH_GENERIC=1 EXP_AMD_CYL_GENERIC=1 python3 tests/fuzz/fuzz_multistep.py 600 443 > $OUT/multistep_generic.txt 2>&1; tail -1 $OUT/multistep_generic.txt
python3 tests/fuzz/fuzz_kdk.py 400 449 > $OUT/kdk.txt 2>&1; tail -1 $OUT/kdk.txt
python3 tests/fuzz/fuzz_store.py 400 457 > $OUT/store.txt 2>&1; tail -1 $OUT/store.txt
bash tools/dbg/prof_r05.sh ${1:-r05f} 2>/dev/null | tail -9
