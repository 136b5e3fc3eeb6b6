#!/bin/bash
# config 4 on ONE stream (EXP_AMD_SIM_OVERLAP=0) under rocprofv3: every kernel's own duration, nothing beside it
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_cfg4_serial$TAG       # TAG=_x ./prof_cfg4_serial.sh --xlist-min -1: extra arguments go to bench_configs.py
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export EXP_AMD_SIM_OVERLAP=0
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/bench_configs.py --only 4 --steps 30 "$@" > $OUT/log.txt 2>&1
python3 - $OUT > $OUT/serial.txt <<'PY'
import csv, glob, sys
rows=[]
for p in glob.glob(sys.argv[1]+"/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ",""), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
rows.sort()
# last master step: from the second-to-last big k_sph_accumulate (> 5e5 threads... the level-0 launch) to the last one
big=[i for i,r in enumerate(rows) if r[2].startswith("k_sph_accumulate") and r[1]-r[0]>200000]
a,b=big[-2],big[-1]
# back up to the key pass that starts the sub-step
while a>0 and not rows[a][2].startswith("k_key_hist<Sph"): a-=1
while b>0 and not rows[b][2].startswith("k_key_hist<Sph"): b-=1
seg=rows[a:b]
print("master step: span %.2f ms, kernel time %.2f ms, %d launches"%((seg[-1][1]-seg[0][0])/1e6, sum(r[1]-r[0] for r in seg)/1e6, len(seg)))
t0=seg[0][0]
for s,e,k,gx,gy,gz in seg:
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  g=({gx},{gy},{gz}) {k[:44]}")
PY
find $OUT -name "*.csv" -size +5M -delete
