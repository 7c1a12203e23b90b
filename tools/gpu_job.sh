#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unigr_gpu.py -q -m gpu 2>&1 | tail -4 | cut -c1-300
timeout 900 python bench.py --mode train_full --steps 20 --warmup 5 --no-cpu-baseline > $O/r2s_train.json 2> $O/r2s_train.err; python3 -c "
import json;d=json.loads([l for l in open('$O/r2s_train.json') if l.startswith('{')][-1]);print('TRAIN',d['value'],d['ms_per_step'],d['verify'])"; tail -3 $O/r2s_train.err
