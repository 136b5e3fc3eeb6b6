#!/bin/bash
# ON THE GPU BOX: the LDS histogram window of the sort passes (SORT_WIN 4096 / 512 / 256 bins: a block of a nearly sorted store touches
# a handful, and zeroes / walks the whole window twice), headline and config 4, variants interleaved.  Build: see profiles/r05_sort_win_ab.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/ab.sh base w512 w256 base w512 w256
for rep in 1 2; do for v in base w512 w256; do
  if [ "$v" = base ]; then unset EXP_AMD_LIB; else export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_$v.so; fi
  echo "[cfg4 $v] $(python3 tools/bench_configs.py --only 4 --steps ${STEPS:-100} 2>&1 | grep -o 'ms_per_master_step[^,]*' | head -1)"
done; done
