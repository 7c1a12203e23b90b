#!/bin/bash
# ragged-row tiles: GEMM tests, then same-board A/B of the library (old = librga3_hip_old.so) on the forward and the training step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu -k "gemm" > $O/r05p_gemm_tests.log 2>&1; echo "gemm tests rc $?"; tail -3 $O/r05p_gemm_tests.log | cut -c1-300
bash tools/gpu_r4_ab.sh forward 2>&1 | tee $O/r05p_ab_forward.log
bash tools/gpu_r4_ab.sh train_full 2>&1 | tee $O/r05p_ab_train.log
