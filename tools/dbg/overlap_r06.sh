#!/bin/bash
# ON THE GPU BOX (one gpurun call): does the HBM-bound sort pass hide under the fp64-bound passes?  The split fused step
# (bench.py --split) against the plain one on the headline workload, interleaved; the aux stream confined to n CUs
# (experimental variant of context.hip, EXP_AMD_AUX_CUS); and a kernel trace of the split run -> timeline.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
OUT=$REPO/gpurun_out/overlap_r06; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-sustained --no-live-traffic"
line() { python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); k = d['roofline']['kernels_ms_per_step']
        print('$1', round(d['value']/1e9,3), 'Gp/s', round(d['ms_per_step'],3), 'ms; kernel sum', round(sum(k.values()),3), {a: round(b,3) for a,b in k.items() if b > 0.05}, 'coef00', d['selfcheck']['coef_00_0'])
"; }
{
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6
for rep in 1 2 3; do
  timeout 300 python3 bench.py $ARGS 2>/dev/null | line "plain      "
  timeout 300 python3 bench.py $ARGS --split 2>/dev/null | line "split      "
done
if [ -f exp_amd/libexp_amd_expt.so ]; then
  for cus in 32 64 128 192; do
    EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_expt.so EXP_AMD_AUX_CUS=$cus timeout 300 python3 bench.py $ARGS --split 2>/dev/null | line "split aux=$cus"
  done
  EXP_AMD_LIB=$REPO/exp_amd/libexp_amd_expt.so timeout 300 python3 bench.py $ARGS --split 2>/dev/null | line "split (expt lib, no mask)"
fi
} > $OUT/ab.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_split -- python3 $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-sustained --no-live-traffic --split > $OUT/trace_split.log 2>&1
python3 $REPO/tools/dump_step_timeline.py $OUT/trace_split 2 > $OUT/timeline_split.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_plain -- python3 $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-sustained --no-live-traffic > $OUT/trace_plain.log 2>&1
python3 $REPO/tools/dump_step_timeline.py $OUT/trace_plain 2 > $OUT/timeline_plain.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/ab.txt
