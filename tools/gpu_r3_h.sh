#!/bin/bash
# functional two-rank run of the training step on the one GPU (gloo, both ranks on device 0): launcher, fresh-batch feed per rank, bucket order, sparse row exchange
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
RGA3_BENCH_SHARE_GPU=1 RGA3_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --mode train_full --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_two_ranks_shared_gpu.json 2> $O/r03_two_ranks_shared_gpu.err; tail -c 900 $O/r03_two_ranks_shared_gpu.json; tail -3 $O/r03_two_ranks_shared_gpu.err
