#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp
for v in "" "--no-prefetch" "" "--no-prefetch"; do python3 $R/bench.py --mode lora_fp8 --steps 3 --warmup 1 --no-cpu-baseline $v > $O/r03_pf8.json 2> $O/r03_pf8.err; python3 -c "
import json;d=json.loads(open('$O/r03_pf8.json').read().strip().splitlines()[-1]);print('FP8 [$v]',d['value'],d['ms_per_step'])"; done
python3 $R/bench.py > $O/r03_bench_default_prefetch.json 2> $O/r03_bench_default_prefetch.err; python3 -c "
import json;d=json.loads(open('$O/r03_bench_default_prefetch.json').read().strip().splitlines()[-1]);print('DEFAULT',d['value'],d['ms_per_step'],d['roofline']['forward_ms_per_step'],d['config']['variants'],d['config']['vision_prefetch'])"
