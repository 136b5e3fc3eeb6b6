#!/bin/bash
# ON THE GPU BOX: the short LDS window + short tiles of the sort passes for (1) one-level sphere stores only, (2) block-multistep full sorts and the
# cylinder as well (EXP_AMD_SORT_DENSE in an experimental build of sph.hip / cyl.hip: exp_amd/libexp_amd_exp.so), configs 3 and 4, interleaved
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_exp.so
for rep in 1 2 3; do for m in 1 2; do
  echo "[cfg4 dense=$m] $(EXP_AMD_SORT_DENSE=$m python3 tools/bench_configs.py --only 4 --steps ${STEPS:-100} 2>&1 | grep -o 'ms_per_master_step[^,]*' | head -1)"
done; done
for rep in 1 2 3; do for m in 1 2; do
  echo "[cfg3 dense=$m] $(EXP_AMD_SORT_DENSE=$m python3 tools/bench_configs.py --only 3 --steps 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1) $(EXP_AMD_SORT_DENSE=$m python3 tools/bench_configs.py --only 3 --steps 30 2>/dev/null | grep -o '"k_scatter_adv": [0-9.]*' | head -1)"
done; done
