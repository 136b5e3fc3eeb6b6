#!/bin/bash
# ON THE GPU BOX: tools/dbg/cross_force.py under library variants, interleaved.  tools/dbg/ab_cross_force.sh base <suffix> ...
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$REPO"
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset EXP_AMD_LIB; else export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_$v.so; fi
  echo "[$v] $(python3 tools/dbg/cross_force.py 24 2>&1 | tail -1)"
done; done
