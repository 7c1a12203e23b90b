#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/tools/hiera_attn_probe.py > $O/r3p_a.txt 2>&1; grep "global" $O/r3p_a.txt | head -2
RGA3_ATTN_96=1 timeout 600 python3 $R/tools/hiera_attn_probe.py > $O/r3p_b.txt 2>&1; grep "global" $O/r3p_b.txt | head -2
