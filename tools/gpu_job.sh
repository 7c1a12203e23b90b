#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_qwen_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu > $O/r3g_tests.log 2>&1; tail -3 $O/r3g_tests.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do
RGA3_ATTN_ROPE_OLD=$v python3 $R/bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r3g_fwd_$v.json 2> $O/r3g_fwd_$v.err; python3 -c "
import json;d=json.loads(open('$O/r3g_fwd_$v.json').read().strip().splitlines()[-1]);print('FWD old_rope=$v',d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('whole_forward_frac'))"
done
