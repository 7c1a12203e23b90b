#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/r02_gpu_tests.log 2>&1; tail -6 $O/r02_gpu_tests.log | cut -c1-300
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r02_bench_headline_final.json 2> $O/r02_bench_headline_final.err; python3 -c "
import json;d=json.loads([l for l in open('$O/r02_bench_headline_final.json') if l.startswith('{')][-1]);print('HEAD',d['value'],d['ms_per_step'],d['roofline']['whole_forward_frac'],d['roofline']['frac'],d['roofline']['forward_ms_per_step'],d['roofline_fwd_bwd']['frac'],d['cpu_baseline']['value'])"; tail -4 $O/r02_bench_headline_final.err
RGA3_BENCH_SHARE_GPU=1 RGA3_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --mode train_full --steps 2 --warmup 1 --no-cpu-baseline > $O/r02_two_ranks_shared_gpu.json 2> $O/r02_two_ranks_shared_gpu.err; tail -c 400 $O/r02_two_ranks_shared_gpu.json
