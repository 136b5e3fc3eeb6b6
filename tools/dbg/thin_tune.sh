#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for cfg in "4 8" "16 32" "32 32" "64 64"; do
  set -- $cfg
  OUT=$REPO/gpurun_out/thin_tp$1
  rm -rf $OUT; mkdir -p $OUT
  EXP_AMD_THIN_MAX=100000 EXP_AMD_THIN_TP=$1 EXP_AMD_THIN_TPA=$2 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/bench_configs.py --only 4 --steps 40 > $OUT/log.txt 2>&1
  echo "=== tp $1 tpa $2: $(grep -o 'ms_per_master_step[^,]*' $OUT/log.txt | head -1)"
  python3 $REPO/tools/dbg/thin_kernel_times.py $OUT | grep thin
  find $OUT -name "*.csv" -delete
done
