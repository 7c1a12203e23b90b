#!/bin/bash
# epilogue rework: kernel tests, K = 576 probe (no PMC), forward bench with variants
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/r04i_kernel_tests.log 2>&1; echo "kernel tests rc $?"; tail -5 $O/r04i_kernel_tests.log
timeout -k 10 300 python3 tools/probes/k576_probe.py time $O/r04_k576_time_i.json > $O/r04_k576_time_i.log 2>&1; cat $O/r04_k576_time_i.log | grep -v amdgpu.ids
timeout -k 10 600 python bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r04i_fwd.json 2> $O/r04i_fwd.err; echo "fwd rc $?"; python3 - <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else '/root/repo/gpurun_out/r04i_fwd.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['roofline'].get('whole_forward_frac'), d['roofline'].get('frac'), d['roofline'].get('variants_ms'))
except Exception as e: print('parse', e)
P
