#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp
for i in 1 2; do
for v in "" "--no-prefetch"; do python3 $R/bench.py --mode train_full --steps 15 --warmup 4 --no-cpu-baseline $v > $O/r03_pf.json 2> $O/r03_pf.err; python3 -c "
import json;d=json.loads(open('$O/r03_pf.json').read().strip().splitlines()[-1]);print('TRAIN [$v]',d['value'],d['ms_per_step'],d['verify']['loss_first_last'])"; tail -1 $O/r03_pf.err | cut -c1-200; done; done
