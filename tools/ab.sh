#!/bin/bash
# ON THE GPU BOX: A/B library variants on bench.py.  tools/ab.sh <suffix> [<suffix> ...]   ("" = default lib)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
for v in "$@"; do
  if [ "$v" = base ]; then unset EXP_AMD_LIB; else export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_$v.so; fi
  echo "== $v"
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-sustained 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); k = d['roofline']['kernels_ms_per_step']
        print(round(d['value']/1e9,3), 'Gp/s', round(d['ms_per_step'],3), 'ms', {a: round(b,2) for a,b in k.items() if b > 0.1})
"
done
