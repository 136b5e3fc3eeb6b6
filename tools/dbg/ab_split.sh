#!/bin/bash
# A/B of the split fused step on the headline workload: bench.py --split (exp_amd_ctx_set_split_min) against the plain step
# (superseded by tools/dbg/overlap_r06.sh, which also traces the timeline)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for rep in 1 2; do
for mode in "" "--split"; do
  echo "== ${mode:-plain}"
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $mode $* 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); k = d['roofline']['kernels_ms_per_step']
        print(round(d['value']/1e9,3), 'Gp/s', round(d['ms_per_step'],3), 'ms; kernels sum', round(sum(k.values()),3), {a: round(b,3) for a,b in k.items()}, d['selfcheck'])
"
done
done
