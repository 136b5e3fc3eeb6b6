#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for v in base prev; do
  if [ "$v" = base ]; then unset EXP_AMD_LIB; else export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_$v.so; fi
  echo "== $v"; timeout 200 python -m pytest tests/test_sph_gpu.py -m gpu -q -k "full_size" 2>&1 | grep -E "^E .*assert np|passed|failed" | head -3
done
