#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
run() {
  echo "== $*"
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); k = d['roofline']['kernels_ms_per_step']
        print(round(d['value']/1e9,3), 'Gp/s', round(d['ms_per_step'],3), 'ms; kernels sum', round(sum(k.values()),3), {a: round(b,3) for a,b in k.items()})
"
}
python -c "
import ctypes
h=ctypes.CDLL('libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo),ctypes.byref(hi)); print('prio range', lo.value, hi.value)"
run EXP_AMD_SPLIT_MIN=0
run EXP_AMD_SPLIT_MIN=4000000
run EXP_AMD_SPLIT_MIN=4000000 EXP_AMD_AUX_PRIO=0
run EXP_AMD_SPLIT_MIN=4000000 EXP_AMD_AUX_PRIO=-1
