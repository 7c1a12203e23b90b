#!/bin/bash
# round 5, sixth GPU call: the LDS-DMA form of the fused stage-2 Hiera MLP: parity, A/B inside the encoder, SAM2-L full-size encoder test
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest -x -q -m gpu tests/test_sam2_kernels_gpu.py -k "hiera" > $O/r05f_hiera.log 2>&1; echo "hiera rc $?"; tail -6 $O/r05f_hiera.log | cut -c1-700
timeout -k 10 600 python3 tools/ab_hiera_mlp.py 5 > $O/r05f_ab_hiera_mlp.log 2>&1; echo "ab rc $?"; grep -v amdgpu $O/r05f_ab_hiera_mlp.log | tail -6
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_fullsize_parity_gpu.py -k "hiera" > $O/r05f_sam2.log 2>&1; echo "sam2 rc $?"; tail -4 $O/r05f_sam2.log | cut -c1-600
