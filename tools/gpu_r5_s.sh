#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 300 python3 tools/probes/ragged_items.py 2>&1 | grep -v amdgpu.ids | tee $O/r05s_ragged_items.log
