#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sam2_gpu.py tests/test_unigr_gpu.py tests/test_train_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" > gpurun_out/r2c_tests.log
tail -60 gpurun_out/r2c_tests.log | cut -c1-400
