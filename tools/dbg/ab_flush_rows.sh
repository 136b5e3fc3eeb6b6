#!/bin/bash
# ON THE GPU BOX: the S6 accumulation at 2 against 3 waves per SIMD (FLUSH_ROWS 16 / 8 / 4 + ACC_P0_LDS 2000: the block's LDS 77.9 / 61 /
# 52.9 KB), configs 2 and 4, variants interleaved on one box.  Build first:
#   tools/build_variant_tu.sh fr8 sph_inst_L6 "-DFLUSH_ROWS=8 -DACC_P0_LDS=2000"; tools/build_variant_tu.sh fr4 sph_inst_L6 "-DFLUSH_ROWS=4 -DACC_P0_LDS=2000"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/ab_cfg.sh 2 base fr8 fr4 base fr8 fr4
for rep in 1 2 3; do for v in base fr8 fr4; do
  if [ "$v" = base ]; then unset EXP_AMD_LIB; else export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_$v.so; fi
  echo "[cfg4 $v] $(python3 tools/bench_configs.py --only 4 --steps ${STEPS:-100} 2>&1 | grep -o 'ms_per_master_step[^,]*' | head -1)"
done; done
