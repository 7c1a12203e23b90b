#!/bin/bash
# multi-rank readiness: the shared-GPU two-rank test, and the 2-rank bench line over gloo on one GPU (functional: comm fields)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_ddp_shared_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -4
RGA3_BENCH_SHARE_GPU=1 RGA3_BENCH_BACKEND=gloo timeout 1200 python3 bench.py --gpus 2 --mode train_full --steps 4 --warmup 2 --no-cpu-baseline > $O/r04_two_ranks_shared_gpu.json 2> $O/r04_two_ranks_shared_gpu.err
tail -3 $O/r04_two_ranks_shared_gpu.err; python3 -c "
import json
d=json.loads(open('$O/r04_two_ranks_shared_gpu.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['n_gpus'], d['comm'])"
