#!/bin/bash
# one gpurun call: round 6's profile set -- as prof_r05.sh without the instrumented accumulate build (headline stats + HBM
# counters + SQ counter sets, configs 2 and 3 with counters, config 4 kernel trace + timeline, the driver's bench command)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06a}
bash $REPO/tools/profile.sh $TAG > $REPO/gpurun_out/prof_${TAG}.log 2>&1
bash $REPO/tools/pmc_multi.sh $TAG "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
     "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
     "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
     "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
     > $REPO/gpurun_out/pmc_${TAG}.log 2>&1
cat $REPO/gpurun_out/pmc_${TAG}_*/summary.txt > $REPO/gpurun_out/pmc_${TAG}_all.txt
bash $REPO/tools/profile_cfg.sh ${TAG}_cfg2 2 8 > $REPO/gpurun_out/prof_${TAG}_cfg2.log 2>&1
bash $REPO/tools/profile_cfg.sh ${TAG}_cfg3 3 8 > $REPO/gpurun_out/prof_${TAG}_cfg3.log 2>&1
bash $REPO/tools/dbg/prof_cfg4_timeline.sh > $REPO/gpurun_out/prof_${TAG}_cfg4.log 2>&1
python3 $REPO/bench.py > $REPO/gpurun_out/bench_${TAG}.json 2> $REPO/gpurun_out/bench_${TAG}.err
tail -3 $REPO/gpurun_out/prof_${TAG}.log; head -8 $REPO/gpurun_out/prof_cfg4/trace.txt; tail -c 1500 $REPO/gpurun_out/bench_${TAG}.json
