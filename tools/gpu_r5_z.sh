#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "round5_tilings or ragged_last or four_wave" > $O/r05z_tests.log 2>&1; echo "rc $?"; tail -4 $O/r05z_tests.log | cut -c1-300
