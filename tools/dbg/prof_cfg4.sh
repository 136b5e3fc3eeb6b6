#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_cfg4${1:+_$1}
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/bench_configs.py --only 4 --steps 40 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
rows=[]
for p in glob.glob(sys.argv[1]+"/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        rows.append((r["Kernel_Name"].split("(")[0].replace("void ",""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r:r[1])
# last 40% of the trace = steady master steps
t0=rows[int(len(rows)*0.6)][1]
sel=[r for r in rows if r[1]>=t0]
busy=sum(e-s for _,s,e in sel); span=sel[-1][2]-sel[0][1]
print("launches",len(sel),"busy ms",busy/1e6,"span ms",span/1e6,"gpu busy frac",busy/span)
agg=collections.defaultdict(list)
for k,s,e in sel: agg[k].append((e-s)/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:14]:
    v2=sorted(v)
    print(f"{k[:40]:40s} n={len(v):5d} total={sum(v)/1e3:8.2f} ms  median={v2[len(v2)//2]:8.1f} us  p90={v2[int(len(v2)*0.9)]:8.1f}  max={v2[-1]:8.1f}")
PY
python3 $REPO/tools/trace_cfg4.py $OUT
find $OUT -name "*.csv" -size +5M -delete
