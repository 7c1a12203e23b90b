#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attn" 2>&1 | tail -15 | cut -c1-300 > gpurun_out/r2j_attn_tests.log; cat gpurun_out/r2j_attn_tests.log
timeout 300 python tools/attn_rates.py > gpurun_out/r2j_attn_rates.log 2>&1; cat gpurun_out/r2j_attn_rates.log
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "fp8" 2>&1 | tail -25 | cut -c1-400
