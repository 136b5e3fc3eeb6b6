#!/bin/bash
# ON THE GPU BOX: socket power and shader clock while the fused step runs plain / split (long regions, sampled every 0.2 s)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
OUT=$REPO/gpurun_out/power_r06; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 600 --warmup 5 --no-cpu-baseline --no-other-configs --no-sustained --no-live-traffic"
probe() {   # $1 = tag, rest = bench args
  tag=$1; shift
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT/smi_$tag.txt &
  SP=$!
  timeout 300 python3 bench.py $ARGS "$@" 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$tag', round(d['ms_per_step'],3), 'ms/step')
" >> $OUT/summary.txt
  kill $SP
  python3 - $OUT/smi_$tag.txt $tag >> $OUT/summary.txt <<'PY'
import re, sys
rows = []
for ln in open(sys.argv[1]):
    p = re.search(r"Power \(W\): ([0-9.]+)", ln); s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", ln)
    if p and s:
        rows.append((float(p.group(1)), int(s.group(1))))
busy = [r for r in rows if r[0] > 0.6 * max(x[0] for x in rows)]
if busy:
    print(sys.argv[2], "samples under load", len(busy), "power W mean/max", round(sum(r[0] for r in busy) / len(busy)), max(r[0] for r in busy),
          "sclk MHz mean/min/max", round(sum(r[1] for r in busy) / len(busy)), min(r[1] for r in busy), max(r[1] for r in busy))
PY
}
rocm-smi --showmaxpower 2>/dev/null | grep -i power >> $OUT/summary.txt
probe plain
probe split --split
cat $OUT/summary.txt
