#!/bin/bash
# round 4, job C: RMSNorm-folded GEMMs -- tests, then the forward bench with the fold on / off on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R && timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_qwen_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu -k "gemm or qwen or fold or vit or decoder or forward or parity or cache or generate" > $O/r04c_tests.log 2>&1; tail -5 $O/r04c_tests.log
for i in 1 2; do
python3 bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold on ', d['ms_per_step'], d['roofline']['whole_forward_frac'], d['roofline']['frac'])"
RGA3_RMS_FOLD=0 python3 bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold off', d['ms_per_step'], d['roofline']['whole_forward_frac'], d['roofline']['frac'])"
done
