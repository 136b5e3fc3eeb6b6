#!/bin/bash
# one gpurun call: the round's profiles (headline, config 3, config 4)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02a}
bash $REPO/tools/profile.sh $TAG > $REPO/gpurun_out/prof_${TAG}.log 2>&1
bash $REPO/tools/profile_cfg.sh ${TAG}_cfg3 3 8 > $REPO/gpurun_out/prof_${TAG}_cfg3.log 2>&1
bash $REPO/tools/dbg/prof_cfg4.sh > $REPO/gpurun_out/prof_${TAG}_cfg4.log 2>&1
tail -3 $REPO/gpurun_out/prof_${TAG}.log; tail -20 $REPO/gpurun_out/prof_${TAG}_cfg4.log
