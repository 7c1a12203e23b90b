#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for gm in 0 2 8 16; do echo "== GROUPM=$gm"; RGA3_GEMM_GROUPM=$gm timeout 600 python3 $R/tools/gemm_tile_probe.py 2>&1 | grep "M=65536\|M=262144\|M=8192" | cut -c1-200; done
