#!/bin/bash
# ON THE GPU BOX: the split fused step with the sort passes' footprint per CU bounded by an LDS request (EXP_AMD_SPLIT_LDS,
# experimental build): do the fp64-bound waves keep the CU then?
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
OUT=$REPO/gpurun_out/overlap_r06b; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-sustained --no-live-traffic"
line() { python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); k = d['roofline']['kernels_ms_per_step']
        print('$1', round(d['value']/1e9,3), 'Gp/s', round(d['ms_per_step'],3), 'ms; kernel sum', round(sum(k.values()),3), {a: round(b,3) for a,b in k.items() if b > 0.05})
"; }
export EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_expt.so
{
timeout 300 python3 bench.py $ARGS 2>/dev/null | line "plain          "
for lds in 0 16384 32768 49152; do
  EXP_AMD_SPLIT_LDS=$lds timeout 300 python3 bench.py $ARGS --split 2>/dev/null | line "split lds=$lds"
done
timeout 300 python3 bench.py $ARGS 2>/dev/null | line "plain          "
EXP_AMD_SPLIT_LDS=49152 timeout 300 python3 bench.py $ARGS --split 2>/dev/null | line "split lds=49152"
EXP_AMD_SPLIT_LDS=32768 timeout 300 python3 bench.py $ARGS --split 2>/dev/null | line "split lds=32768"
} > $OUT/ab.txt 2>&1
cd /tmp; export TMPDIR=/tmp
EXP_AMD_SPLIT_LDS=49152 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_split -- python3 $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-sustained --no-live-traffic --split > $OUT/trace_split.log 2>&1
python3 $REPO/tools/dump_step_timeline.py $OUT/trace_split 2 > $OUT/timeline_split_lds48k.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/ab.txt
