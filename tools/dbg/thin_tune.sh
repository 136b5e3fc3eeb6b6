#!/bin/bash
# (needs make EXPERIMENTAL=1 for EXP_AMD_THIN_V / EXP_AMD_THIN_TILE; EXP_AMD_THIN_MAX is no longer read -- the thin path's
# bound is exp_amd_ctx_set_thin_max, 8192 by default)
# rocprofv3 kernel durations of the thin kernels by grid size, for the settings given as "V TILE" pairs
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  OUT=$REPO/gpurun_out/thin_v$1_$2
  rm -rf $OUT; mkdir -p $OUT
  EXP_AMD_THIN_MAX=16384 EXP_AMD_THIN_V=$1 EXP_AMD_THIN_TILE=$2 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/bench_configs.py --only 4 --steps 40 > $OUT/log.txt 2>&1
  echo "=== v $1 tile $2: $(grep -o 'ms_per_master_step[^,]*' $OUT/log.txt | head -1)"
  python3 $REPO/tools/dbg/thin_kernel_times.py $OUT | grep "thin\|wave\|tile"
  find $OUT -name "*.csv" -delete
done
