#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_sam2_gpu.py -x -q -m gpu > $O/r3a_tests.log 2>&1; tail -5 $O/r3a_tests.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/tools/gemm_shape_table.py $O/r3a_gemm_shapes.json > $O/r3a_gemm_shapes.txt 2>&1; grep -v amdgpu.ids $O/r3a_gemm_shapes.txt | head -45 | cut -c1-160
python3 $R/bench.py --mode train_full --steps 10 --warmup 3 --no-cpu-baseline > $O/r3a_train.json 2> $O/r3a_train.err; python3 -c "
import json;d=json.loads(open('$O/r3a_train.json').read().strip().splitlines()[-1]);print('TRAIN',d['value'],d['ms_per_step'])"
