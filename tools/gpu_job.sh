#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_qwen_gpu.py -x -q -m gpu -k "cross_entropy or training_step or full_forward or loss" > $O/r3t_tests.log 2>&1; tail -3 $O/r3t_tests.log | cut -c1-200
