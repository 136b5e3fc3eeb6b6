#!/bin/bash
# ON THE GPU BOX: several separate --pmc passes over bench.py.  tools/pmc_multi.sh <tag> "<set1>" "<set2>" ...
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
i=0
for CTRS in "$@"; do
  i=$((i+1))
  timeout 300 bash "$REPO/tools/pmc.sh" "${TAG}_$i" "$CTRS" | grep -E "k_sph_force|k_sph_accumulate"
done
