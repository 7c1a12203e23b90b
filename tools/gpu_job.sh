#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_folded or gemm_plain" > $O/r3l_tests.log 2>&1; tail -6 $O/r3l_tests.log | cut -c1-250
timeout 1500 python -m pytest tests/test_sam2_gpu.py tests/test_sam2_kernels_gpu.py tests/test_fullsize_parity_gpu.py tests/test_unigr_gpu.py -x -q -m gpu > $O/r3l_tests2.log 2>&1; tail -4 $O/r3l_tests2.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do RGA3_LN_FOLD=$v NF=16 timeout 600 python3 $R/tools/sam2_encoder_probe.py 8 2>&1 | grep "ms per" | sed "s/^/LN_FOLD=$v /"; done
python3 $R/bench.py --mode train_full --steps 10 --warmup 3 --no-cpu-baseline > $O/r3l_train.json 2> $O/r3l_train.err; python3 -c "
import json;d=json.loads(open('$O/r3l_train.json').read().strip().splitlines()[-1]);print('TRAIN',d['value'],d['ms_per_step'])"
RGA3_LN_FOLD=0 python3 $R/bench.py --mode train_full --steps 10 --warmup 3 --no-cpu-baseline > $O/r3l_train0.json 2> $O/r3l_train0.err; python3 -c "
import json;d=json.loads(open('$O/r3l_train0.json').read().strip().splitlines()[-1]);print('TRAIN no-fold',d['value'],d['ms_per_step'])"
