#!/bin/bash
# round 5: the driver's two commands on the current tree (full GPU suite, smoke), then every bench mode once
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 2400 python3 -m pytest tests/ -x -q -m gpu > $O/r05g_gpu_tests.log 2>&1; echo "suite rc $?"; tail -4 $O/r05g_gpu_tests.log | cut -c1-800
timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r05g_smoke.log 2>&1; echo "smoke rc $?"; tail -3 $O/r05g_smoke.log
timeout -k 10 600 python3 tools/ab_hiera_mlp.py 5 > $O/r05g_ab_hiera_mlp.log 2>&1; grep -v amdgpu $O/r05g_ab_hiera_mlp.log | tail -3
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $O/r05g_headline.json 2> $O/r05g_headline.err; echo "headline rc $?"
python3 - $O/r05g_headline.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('HEAD', d['value'], d['ms_per_step'], 'fwd', r.get('forward_ms_per_step'), r.get('whole_forward_frac'), r.get('frac'), 'traffic', r.get('traffic'), r.get('traffic_stale'), d['verify'].get('stream_k_timeouts'), d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
except Exception as e: print('parse', e)
P
timeout -k 10 900 python3 bench.py --mode lora_fp8 --steps 3 --warmup 1 --no-cpu-baseline > $O/r05g_fp8.json 2> $O/r05g_fp8.err; echo "fp8 rc $?"
python3 - $O/r05g_fp8.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('FP8', d['value'], d['ms_per_step'], r.get('frac'), (r.get('gemm_family') or {}).get('by_arithmetic'))
except Exception as e: print('parse', e)
P
