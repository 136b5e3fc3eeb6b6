#!/bin/bash
# ON THE DEV BOX, after `gpurun -- bash tools/dbg/prof_r02.sh <tag>`: condense gpurun_out/ into the
# committed summaries profiles/<round>_*  (round = tag without its trailing letter, e.g. r02c -> r02).
#   tools/publish_profiles.sh r02c
set -e
cd "$(dirname "$0")/.."
TAG=$1; RND=${TAG%[a-z]}
# gpurun MERGES a call's files into gpurun_out/: a tag used before leaves its old traces beside the new ones and the
# summaries would average both.  Keep only what the newest stats run (and anything within 30 minutes of it) wrote.
NEWEST=$(ls -t gpurun_out/prof_$TAG/stats.log gpurun_out/prof_$TAG/stats/*/*.csv 2>/dev/null | head -1)
if [ -n "$NEWEST" ]; then
  REF=$(mktemp); touch -d "$(date -r "$NEWEST" '+%Y-%m-%d %H:%M:%S') 30 minutes ago" "$REF"
  find gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_cfg2 gpurun_out/prof_${TAG}_cfg3 gpurun_out/prof_cfg4 gpurun_out/pmc_${TAG}_* -type f ! -newer "$REF" -delete 2>/dev/null || true
  rm -f "$REF"
fi
python3 tools/summarize_profile.py gpurun_out/prof_$TAG profiles/$RND > /dev/null
cp gpurun_out/pmc_${TAG}_all.txt profiles/${RND}_pmc.txt
sed -i "1i # rocprofv3 --pmc passes over 'python bench.py --steps 3 --warmup 1' (1e8 NFW, S10), one counter set per pass (tools/pmc_multi.sh); per-launch means, first launch dropped" profiles/${RND}_pmc.txt
[ -d gpurun_out/prof_${TAG}_cfg2 ] && python3 tools/summarize_cfg.py gpurun_out/prof_${TAG}_cfg2 profiles/${RND}_cfg2 > /dev/null
python3 tools/summarize_cfg.py gpurun_out/prof_${TAG}_cfg3 profiles/${RND}_cfg3 > /dev/null
if [ -f gpurun_out/prof_cfg4/trace.txt ]; then
  # (tools/dbg/prof_cfg4_timeline.sh made the summaries on the GPU box; the raw trace was too large to come back)
  python3 - <<PY
import sys
sys.argv = ["x"]
sys.path.insert(0, "tools")
import glob, summarize_cfg
summarize_cfg.stats(sorted(glob.glob("gpurun_out/prof_cfg4/*/*_kernel_stats.csv"))[-1], "profiles/${RND}_cfg4")
PY
  cp gpurun_out/prof_cfg4/trace.txt profiles/${RND}_cfg4_trace.txt
  cp gpurun_out/prof_cfg4/timeline.txt profiles/${RND}_cfg4_timeline.txt
else
  python3 tools/summarize_cfg.py --cfg4 gpurun_out/prof_cfg4 profiles/${RND}_cfg4 > /dev/null
fi
grep "^{" gpurun_out/prof_$TAG/stats.log | tail -1 > profiles/${RND}_bench_under_rocprof.json
ls -la profiles/${RND}_*
