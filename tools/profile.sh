#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + HBM traffic counters for bench.py.
#   tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,fetch,write}/ ; summaries are copied into profiles/ by
# tools/summarize_profile.py afterwards (on the dev box).
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-sustained $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/write.log" 2>&1
# keep only the small CSVs (kernel stats + counter collection); traces can be large
find "$OUT" -name "*.csv" -size +20M -delete
ls -R "$OUT" | head -50
tail -2 "$OUT/stats.log"
