#!/bin/bash
# round 5, second GPU call: re-run of the failed new tests (full-depth with the bf16-storage yardstick, LoRA fold fallback), multi-object SAM2 tests + bench, rest of the suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 1500 python3 -m pytest -x -q -s -m gpu tests/test_fulldepth_parity_gpu.py > $O/r05b_fulldepth.log 2>&1; echo "fulldepth rc $?"; grep -E "FULL_DEPTH_7B|FP8_LAYER|passed|failed|Error|assert" $O/r05b_fulldepth.log | cut -c1-1800 | tail -12
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_sam2_gpu.py > $O/r05b_sam2.log 2>&1; echo "sam2 rc $?"; tail -15 $O/r05b_sam2.log | cut -c1-600
for n in 1 2 4; do
timeout -k 10 900 python3 bench.py --mode sam2_stream --objects $n --steps 5 --warmup 2 --no-cpu-baseline > $O/r05b_stream_$n.json 2> $O/r05b_stream_$n.err; echo "stream $n rc $?"
python3 - $O/r05b_stream_$n.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('STREAM', d['value'], d['ms_per_step'], d['config'].get('multi_object'), d['roofline'].get('frac'))
except Exception as e: print('parse', e)
P
done
timeout -k 10 2400 python3 -m pytest tests/ -x -q -m gpu --deselect tests/test_fulldepth_parity_gpu.py --deselect tests/test_sam2_gpu.py > $O/r05b_gpu_tests.log 2>&1; echo "suite rc $?"; tail -12 $O/r05b_gpu_tests.log | cut -c1-800
