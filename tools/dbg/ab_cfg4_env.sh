#!/bin/bash
# ON THE GPU BOX: config 4 under alternating environment settings, same box, interleaved.  tools/dbg/ab_cfg4_env.sh "A=1" "B=2" ...
# ("" = the default)
for rep in 1 2; do for e in "$@"; do
  echo "[$e] $(env $e python3 tools/bench_configs.py --only 4 --steps ${STEPS:-200} 2>&1 | grep -o 'ms_per_master_step[^,]*' | head -1)"
done; done
