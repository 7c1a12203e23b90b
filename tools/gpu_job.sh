#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { timeout 600 python3 $R/tools/attn_rates.py 2>&1 | grep -v amdgpu | tail -7 | cut -c1-90; timeout 600 python3 $R/tools/hiera_attn_probe.py 2>&1 | grep "global.*packed qkv  impl=0" | head -1; }
echo "== new (one barrier)"; run
cp $R/tools/_ab/attn_fwd_old.hip $R/rga3-release_amd/csrc/attn_fwd.hip; make -C $R/rga3-release_amd/csrc > /dev/null 2>&1
echo "== old (two barriers)"; run
cp $R/tools/_ab/attn_fwd_new.hip $R/rga3-release_amd/csrc/attn_fwd.hip; make -C $R/rga3-release_amd/csrc > /dev/null 2>&1
echo "== new again"; run
