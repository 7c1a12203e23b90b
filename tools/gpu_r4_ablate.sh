#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/rga3-release_amd/csrc && rm -f build/attn_causal32.o && make AB=1 -j8 2>&1 | grep -E "error|librga3" | head -3
cd $R && python3 tools/probes/causal32_ablate.py 2>&1 | grep -v amdgpu
