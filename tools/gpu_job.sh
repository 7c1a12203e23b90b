#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -s -k "swiglu or attn" 2>&1 | grep "SWIGLU\|passed\|failed\|^E  " | cut -c1-300
timeout 300 python tools/attn_rates.py 2>&1 | grep -v amdgpu.ids
