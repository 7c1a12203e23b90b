#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for s in - abl1 abl2 abl3 abl4; do timeout -k 10 300 python3 tools/probes/ragged_cost.py $s 2>&1 | grep -v amdgpu.ids; done | tee $O/r05r_ragged_cost.log
