#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_sam2_gpu.py tests/test_unigr_gpu.py tests/test_fullsize_parity_gpu.py tests/test_qwen_gpu.py -x -q -m gpu > $O/r3c_tests.log 2>&1; tail -5 $O/r3c_tests.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --mode train_full --steps 10 --warmup 3 --no-cpu-baseline > $O/r3c_train.json 2> $O/r3c_train.err; python3 -c "
import json;d=json.loads(open('$O/r3c_train.json').read().strip().splitlines()[-1]);print('TRAIN',d['value'],d['ms_per_step'])"
python3 $R/bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r3c_fwd.json 2> $O/r3c_fwd.err; python3 -c "
import json;d=json.loads(open('$O/r3c_fwd.json').read().strip().splitlines()[-1]);print('FWD',d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('whole_forward_frac'))"
rm -rf /tmp/pt; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o tr -- python3 $R/bench.py --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r3c_trace.log 2>&1
python3 $R/tools/step_timeline.py /tmp/pt --bin-ms 10 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r3c_nongemm.txt 2>&1; tail -n +20 $O/r3c_nongemm.txt | head -36 | cut -c1-150
