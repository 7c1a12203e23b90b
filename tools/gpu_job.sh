#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $O/r3k_tests.log 2>&1; tail -3 $O/r3k_tests.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/tools/gemm_tile_probe.py > $O/r3k_tiles.txt 2>&1; grep -v amdgpu.ids $O/r3k_tiles.txt | grep "M=2112" | cut -c1-220
