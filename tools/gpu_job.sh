#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/tools/gemm_tile_probe.py > $O/r3m_tiles.txt 2>&1; grep -v amdgpu.ids $O/r3m_tiles.txt | cut -c1-230
