#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R; timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_qwen_gpu.py tests/test_fullsize_parity_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/r4g_tests.log 2>&1; tail -2 $O/r4g_tests.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r4g_fwd.json 2> $O/r4g_fwd.err; python3 -c "
import json;d=json.loads(open('$O/r4g_fwd.json').read().strip().splitlines()[-1]);print('FWD',d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('whole_forward_frac'))"
python3 $R/bench.py --mode train_full --steps 10 --warmup 3 --no-cpu-baseline > $O/r4g_train.json 2> $O/r4g_train.err; python3 -c "
import json;d=json.loads(open('$O/r4g_train.json').read().strip().splitlines()[-1]);print('TRAIN',d['value'],d['ms_per_step'])"
