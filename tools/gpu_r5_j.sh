#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 -m pytest -x -q -m gpu tests/test_kernels_gpu.py -k "layernorm" tests/test_sam2_kernels_gpu.py > $O/r05j_ln.log 2>&1; echo "ln tests rc $?"; tail -3 $O/r05j_ln.log | cut -c1-500
for v in oldattn -; do echo "lib $v:"; timeout -k 10 300 python3 tools/probes/run_with_lib.py $v tools/probes/ln_stats_probe.py 2>&1 | grep rows; done
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_fullsize_parity_gpu.py -k "hiera" 2>&1 | tail -2
for rep in 1 2; do for v in oldattn -; do
  timeout -k 10 600 python3 tools/probes/bench_with_lib.py $v --mode train_full --steps 20 --warmup 5 --no-cpu-baseline > $O/r05j_ab_${v}_$rep.json 2> $O/r05j_ab_${v}_$rep.err
  python3 - $O/r05j_ab_${v}_$rep.json $v <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('train_full', sys.argv[2], d['value'], d['ms_per_step'])
except Exception as e: print('parse', e)
P
done; done
