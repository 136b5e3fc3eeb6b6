#!/bin/bash
# ON THE GPU BOX: separate rocprofv3 --pmc passes over a python script, per-kernel means of the kernels matching FILTER.
#   FILTER=k_sph_force tools/dbg/pmc_script.sh <tag> <script.py> "<args>" "<set1>" "<set2>" ...
set -u
TAG=$1; SCRIPT=$2; ARGS=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
i=0
for CTRS in "$@"; do
  i=$((i+1))
  OUT=$REPO/gpurun_out/pmcs_${TAG}_$i
  mkdir -p "$OUT"
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/$SCRIPT" $ARGS > "$OUT/log.txt" 2>&1)
  python3 - "$OUT" "${FILTER:-k_}" <<'PY'
import csv, glob, sys, collections
out, flt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if flt in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k + " : " + ", ".join(f"{c}={sum(v[1:])/max(1,len(v[1:])):.4g}" for c, v in sorted(d.items())))
PY
  find "$OUT" -name "*.csv" -size +5M -delete
done
