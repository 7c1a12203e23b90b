#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 2400 python3 -m pytest tests/ -x -q -m gpu > $O/r05_gpu_tests.log 2>&1; echo "suite rc $?"; tail -2 $O/r05_gpu_tests.log | cut -c1-300
timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/r05_smoke.log
