#!/bin/bash
# round 5, fourth GPU call: the three-phase 192-row GEMM loop (tiles 31 / 32): race screen, per-tile rates, kernel tests; ViT e4m3 test; forward bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python3 tools/probes/tile31_screen.py 30 > $O/r05d_screen.log 2>&1; echo "screen rc $?"; grep -v amdgpu $O/r05d_screen.log | tail -20
timeout -k 10 900 python3 tools/gemm_tile_probe.py > $O/r05d_tile_probe.log 2>&1; echo "probe rc $?"; grep -v amdgpu $O/r05d_tile_probe.log | tail -20
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_kernels_gpu.py -k "gemm" > $O/r05d_gemm_tests.log 2>&1; echo "gemm tests rc $?"; tail -4 $O/r05d_gemm_tests.log | cut -c1-600
timeout -k 10 900 python3 -m pytest -x -q -s -m gpu tests/test_fullsize_parity_gpu.py -k "vit" > $O/r05d_vit.log 2>&1; echo "vit rc $?"; grep -E "VIT_FP8|passed|failed" $O/r05d_vit.log | cut -c1-800
timeout -k 10 900 python3 bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r05d_forward.json 2> $O/r05d_forward.err; echo "forward rc $?"
python3 - $O/r05d_forward.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('FWD', d['value'], d['ms_per_step'], 'whole', r.get('whole_forward_frac'), 'gemm', r.get('frac'), r.get('gemm_ms_per_step'))
except Exception as e: print('parse', e)
P
grep -i "tuner" $O/r05d_forward.err | tail -3
