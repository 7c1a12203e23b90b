#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 tools/probes/forward_aten_census.py > $O/r05aa_census.log 2>&1; echo rc $?; grep -v amdgpu.ids $O/r05aa_census.log | tail -45 | cut -c1-220
