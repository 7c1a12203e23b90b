#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
bash tools/gpu_r4_ab.sh forward 2>&1 | tee $O/r05q_ab_forward.log
bash tools/gpu_r4_ab.sh train_full 2>&1 | tee $O/r05q_ab_train.log
grep -h "tuner" $O/ab_-_2.err | tail -3
