#!/bin/bash
# rocprofv3 kernel trace of config 4 -> trace summary + kernel-by-kernel timeline of one settled master step
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_cfg4
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/bench_configs.py --only 4 --steps 40 > $OUT/log.txt 2>&1
python3 $REPO/tools/trace_cfg4.py $OUT > $OUT/trace.txt 2>&1
python3 $REPO/tools/dump_cfg4_timeline.py $OUT 2 > $OUT/timeline.txt 2>&1
find $OUT -name "*.csv" -size +5M -delete
