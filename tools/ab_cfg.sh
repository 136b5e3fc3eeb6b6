#!/bin/bash
# ON THE GPU BOX: A/B library variants on one secondary configuration.  tools/ab_cfg.sh <config> <suffix> [<suffix> ...]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
CFG=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset EXP_AMD_LIB; else export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_$v.so; fi
  echo "== $v"
  timeout 300 python tools/bench_configs.py --only $CFG --steps 30 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print(round(d.get('ms_per_step', d.get('ms_per_master_step', 0)), 4), 'ms', {a: round(b, 3) for a, b in d.get('kernels_ms_per_step', d.get('kernels_ms_per_master_step', {})).items() if b > 0.02})
"
done
