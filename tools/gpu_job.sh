#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_unigr_gpu.py tests/test_sam2_gpu.py -x -q -m gpu > $O/r3x_tests.log 2>&1; tail -6 $O/r3x_tests.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do RGA3_SAM_OVERLAP=$v python3 $R/bench.py --mode train_full --steps 10 --warmup 3 --no-cpu-baseline > $O/r3x_train_$v.json 2> $O/r3x_train_$v.err; python3 -c "
import json;d=json.loads(open('$O/r3x_train_$v.json').read().strip().splitlines()[-1]);print('TRAIN overlap=$v',d['value'],d['ms_per_step'],d['verify'])"; done
