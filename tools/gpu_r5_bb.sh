#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 400 python3 tools/probes/power_clock_probe.py > $O/r05bb_power.log 2>&1; echo rc $?; grep -v amdgpu.ids $O/r05bb_power.log | tail -14 | cut -c1-400
