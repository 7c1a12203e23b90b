#!/bin/bash
# round 5, first GPU call: the new parity tests (full-depth 7B forward, fp8 layer at S = 4160, give-up poisoning, ADVICE fixes), the forward + headline lines on this
# tree, then the whole GPU suite.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
nproc; free -g | head -2
timeout -k 10 1500 python3 -m pytest -x -q -s -m gpu tests/test_fulldepth_parity_gpu.py > $O/r05a_fulldepth.log 2>&1; echo "fulldepth rc $?"; grep -E "FULL_DEPTH_7B|FP8_LAYER|passed|failed|Error|assert" $O/r05a_fulldepth.log | cut -c1-1500 | tail -12
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_train_gpu.py -k "lora_fold or give_up or health" tests/test_kernels_gpu.py -k "lora_fold or give_up or health or output_view or fp8_quant" > $O/r05a_new.log 2>&1; echo "new tests rc $?"; tail -5 $O/r05a_new.log
timeout -k 10 900 python3 -m pytest -x -q -m gpu "tests/test_fullsize_parity_gpu.py::test_vit_blocks_7b_grid_8_32_32" > $O/r05a_vit.log 2>&1; echo "vit rc $?"; tail -3 $O/r05a_vit.log
timeout -k 10 900 python3 bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r05a_forward.json 2> $O/r05a_forward.err; echo "forward rc $?"
python3 - $O/r05a_forward.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('FWD', d['value'], d['ms_per_step'], 'whole', r.get('whole_forward_frac'), 'gemm', r.get('frac'), r.get('gemm_ms_per_step'), r.get('variants_ms'), 'traffic', r.get('traffic'), r.get('traffic_stale'))
except Exception as e: print('parse', e)
P
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05a_headline.json 2> $O/r05a_headline.err; echo "headline rc $?"
python3 - $O/r05a_headline.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('HEAD', d['value'], d['ms_per_step'], 'fwd', r.get('forward_ms_per_step'), r.get('whole_forward_frac'), r.get('frac'), d['verify'])
except Exception as e: print('parse', e)
P
timeout -k 10 1500 python3 -m pytest tests/ -x -q -m gpu > $O/r05a_gpu_tests.log 2>&1; echo "suite rc $?"; tail -3 $O/r05a_gpu_tests.log
