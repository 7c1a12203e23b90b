#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 tools/probes/w4_probe.py ${1:-5} 2>&1 | grep -v amdgpu.ids | tee $O/r05t_w4_probe.log | cut -c1-330
