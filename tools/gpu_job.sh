#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_unigr_gpu.py tests/test_sam2_kernels_gpu.py -x -q -m gpu > $O/r3y_tests.log 2>&1; tail -6 $O/r3y_tests.log | cut -c1-250
