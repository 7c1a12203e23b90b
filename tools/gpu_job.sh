#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "folded" > $O/r3z_tests.log 2>&1; tail -6 $O/r3z_tests.log | cut -c1-250
timeout 1500 python -m pytest tests/test_qwen_gpu.py tests/test_fullsize_parity_gpu.py tests/test_unigr_gpu.py tests/test_sam2_gpu.py -x -q -m gpu > $O/r3z_tests2.log 2>&1; tail -6 $O/r3z_tests2.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do RGA3_RMS_FOLD=$v python3 $R/bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r3z_fwd_$v.json 2> $O/r3z_fwd_$v.err; python3 -c "
import json;d=json.loads(open('$O/r3z_fwd_$v.json').read().strip().splitlines()[-1]);print('FWD fold=$v',d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('whole_forward_frac'),d['verify']['gemm_worst_rel_l2_vs_tile10'])"; done
