#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attn" 2>&1 | tail -15 | cut -c1-300 > gpurun_out/r2i_attn_tests.log; cat gpurun_out/r2i_attn_tests.log
timeout 300 python tools/attn_rates.py > gpurun_out/r2i_attn_rates.log 2>&1; cat gpurun_out/r2i_attn_rates.log
timeout 1500 python -m pytest tests/test_unigr_gpu.py tests/test_qwen_gpu.py tests/test_fullsize_parity_gpu.py tests/test_sam2_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" > gpurun_out/r2i_tests.log
grep -n "EMU_ERRS\|^E  \|FAILED\|passed\|failed" gpurun_out/r2i_tests.log | cut -c1-600 | tail -30
