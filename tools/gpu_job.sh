#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/r02_gpu_tests.log 2>&1; tail -6 $O/r02_gpu_tests.log | cut -c1-300
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
timeout 900 python bench.py --mode lora_fp8 --steps 3 --warmup 1 > $O/r02_bench_lora_fp8_fused.json 2> $O/r02_bench_lora_fp8_fused.err; tail -c 500 $O/r02_bench_lora_fp8_fused.json
