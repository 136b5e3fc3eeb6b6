#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for v in base prev; do
  if [ "$v" = base ]; then unset EXP_AMD_LIB; else export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_$v.so; fi
  echo "== $v"
  timeout 300 bash tools/pmc.sh ab_$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" | grep -E "k_sph_accumulate|k_sph_force<10, true"
done
