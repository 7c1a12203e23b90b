#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu -k "gemm" > $O/r05u_gemm_tests.log 2>&1; echo "gemm tests rc $?"; tail -3 $O/r05u_gemm_tests.log | cut -c1-300
RGA3_TUNE_SAVE=$O/r05u_tuner_forward.json python3 bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r05u_bench_forward.json 2> $O/r05u_bench_forward.err; tail -c 700 $O/r05u_bench_forward.json; cat $O/r05u_tuner_forward.json; tail -2 $O/r05u_bench_forward.err
