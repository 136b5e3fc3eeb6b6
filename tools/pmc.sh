#!/bin/bash
# ON THE GPU BOX: collect SQ counters for bench.py (one pass).  tools/pmc.sh <tag> "<counters>" [bench args]
set -u
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-sustained "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as f:
    for k, d in sorted(agg.items()):
        line = k + " : " + ", ".join(f"{c}={sum(v[1:])/max(1,len(v[1:])):.4g}" for c, v in sorted(d.items()))
        print(line); f.write(line + "\n")
PY
find "$OUT" -name "*.csv" -size +5M -delete
