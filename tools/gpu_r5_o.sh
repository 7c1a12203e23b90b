#!/bin/bash
# tile 26 (ragged last tile row): race / correctness screen + rates
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 tools/probes/ragged_probe.py ${1:-both} ${2:-20} > $O/r05o_ragged_probe.log 2>&1; echo "probe rc $?"; grep -v amdgpu.ids $O/r05o_ragged_probe.log | cut -c1-400 | tail -45
