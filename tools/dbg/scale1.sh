#!/bin/bash
# emulate the per-GPU share of an 8-GPU strong-scaling run on one GPU (with the all-reduce forced)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for n in 1.25e7 2.5e7 5e7; do
  echo "== nbodies $n"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-sustained --nbodies $n --force-comm 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); k = d['roofline']['kernels_ms_per_step']
        print(round(d['value']/1e9,3), 'Gp/s', round(d['ms_per_step'],3), 'ms; kernels sum', round(sum(k.values()),3), {a: round(b,3) for a,b in k.items()})
"
done
