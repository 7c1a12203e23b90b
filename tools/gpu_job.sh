#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_qwen_gpu.py -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
timeout 300 python tools/attn_rates.py 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r2m_fwd.json 2> $O/r2m_fwd.err; python3 -c "
import json;d=json.loads(open('$O/r2m_fwd.json').read().strip().splitlines()[-1]);print('FWD',d['ms_per_step'],d['roofline']['whole_forward_frac'],d['roofline']['frac'],d['roofline']['gemm_ms_per_step'])"
RGA3_BENCH_SHARE_GPU=1 RGA3_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --mode train_full --steps 2 --warmup 1 --no-cpu-baseline > $O/r2m_two_ranks.json 2> $O/r2m_two_ranks.err; tail -c 1500 $O/r2m_two_ranks.json; tail -12 $O/r2m_two_ranks.err | cut -c1-300
