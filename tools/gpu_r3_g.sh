#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && timeout 600 python3 -m pytest tests/test_sam2_kernels_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/r03_g_tests.log 2>&1; tail -3 $O/r03_g_tests.log
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
timeout 900 $B --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_headline_final_a.json 2> $O/r03_headline_final_a.err; python3 -c "
import json;d=json.loads(open('$O/r03_headline_final_a.json').read().strip().splitlines()[-1]);print('HEADLINE',d['value'],d['ms_per_step'],d['roofline']['forward_ms_per_step'],d['roofline']['whole_forward_frac'],d['config']['variants'])"
export RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o tr -- $B --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r03_prof_train_trace.log 2>&1
python3 $R/tools/step_timeline.py /tmp/pt --bin-ms 5 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r03_train_step_timeline.txt 2>&1; head -3 $O/r03_train_step_timeline.txt | cut -c1-120
unset RGA3_BENCH_TIMED_ONLY
timeout 600 python3 $R/tools/gemm_shape_table.py $O/r03_train_gemm_shapes.json > $O/r03_train_gemm_shapes.txt 2>&1; head -4 $O/r03_train_gemm_shapes.txt | grep -v amdgpu
