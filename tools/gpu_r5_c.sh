#!/bin/bash
# round 5, third GPU call: fused Hiera MLP at C = 288 (parity + A/B), multi-object SAM2 tests + stream bench, fp8 layer leg with its yardstick, LoRA fold fallback
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest -x -q -m gpu tests/test_sam2_kernels_gpu.py -k "hiera" > $O/r05c_hiera.log 2>&1; echo "hiera rc $?"; tail -4 $O/r05c_hiera.log | cut -c1-600
timeout -k 10 600 python3 tools/ab_hiera_mlp.py 5 > $O/r05c_ab_hiera_mlp.log 2>&1; echo "ab rc $?"; grep -v amdgpu $O/r05c_ab_hiera_mlp.log | tail -6
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_sam2_gpu.py tests/test_fullsize_parity_gpu.py -k "sam2 or two_objects or prompt or propagation or hiera or encoder" > $O/r05c_sam2.log 2>&1; echo "sam2 rc $?"; tail -6 $O/r05c_sam2.log | cut -c1-600
timeout -k 10 900 python3 -m pytest -x -q -s -m gpu tests/test_fulldepth_parity_gpu.py -k fp8 > $O/r05c_fp8layer.log 2>&1; echo "fp8 layer rc $?"; grep -E "FP8_LAYER|passed|failed" $O/r05c_fp8layer.log | cut -c1-1500
timeout -k 10 600 python3 -m pytest -x -q -m gpu tests/test_train_gpu.py -k "lora_fold" > $O/r05c_lora.log 2>&1; echo "lora rc $?"; tail -3 $O/r05c_lora.log | cut -c1-800
for n in 1 2 4; do
timeout -k 10 900 python3 bench.py --mode sam2_stream --objects $n --steps 5 --warmup 2 --no-cpu-baseline > $O/r05c_stream_$n.json 2> $O/r05c_stream_$n.err; echo "stream $n rc $?"
python3 - $O/r05c_stream_$n.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('STREAM', d['value'], d['ms_per_step'], d['config'].get('multi_object'), d['roofline'].get('frac'))
except Exception as e: print('parse', e)
P
done
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05c_headline.json 2> $O/r05c_headline.err; echo "headline rc $?"
python3 - $O/r05c_headline.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('HEAD', d['value'], d['ms_per_step'], 'fwd', r.get('forward_ms_per_step'), r.get('whole_forward_frac'), r.get('frac'))
except Exception as e: print('parse', e)
P
