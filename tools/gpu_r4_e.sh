#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R && timeout 900 python3 -m pytest tests/test_qwen_gpu.py tests/test_fullsize_parity_gpu.py tests/test_unigr_gpu.py -x -q -m gpu > $O/r04e_tests.log 2>&1; tail -5 $O/r04e_tests.log
bash tools/gpu_r4_c2.sh
