#!/bin/bash
# ON THE GPU BOX: rocprofv3 over one secondary configuration (tools/bench_configs.py --only N):
# kernel stats, FETCH_SIZE / WRITE_SIZE (separate --pmc passes) and SQ counter sets.
#   tools/profile_cfg.sh <tag> <config> [steps]
# Writes gpurun_out/prof_<tag>/{stats,fetch,write,sq1..}/ and a text summary gpurun_out/prof_<tag>/summary.txt
set -u
TAG=$1; CFG=$2; STEPS=${3:-8}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/bench_configs.py --only $CFG --steps $STEPS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
            "TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum" "TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/sq$i" -- $CMD > "$OUT/sq$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/summary.txt", "w") as f:
    for sub in sorted(glob.glob(out + "/*/")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for path in glob.glob(sub + "*/*_counter_collection.csv"):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith("k_"):
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in sorted(agg.items()):
            line = k + " : " + ", ".join(f"{c}={sum(v[1:])/max(1,len(v[1:])):.5g} (n={len(v)})" for c, v in sorted(d.items()))
            print(line); f.write(line + "\n")
PY
find "$OUT" -name "*.csv" -size +5M -delete
tail -3 "$OUT/stats.log"
