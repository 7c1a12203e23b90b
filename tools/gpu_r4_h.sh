#!/bin/bash
# training-step shape table + step timeline of the current tree (what to work on next), SAM2 / kernel tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_sam2_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "hiera or sam2 or layernorm or gemm" 2>&1 | tail -2
python3 tools/gemm_shape_table.py $O/r04_train_gemm_shapes.json > $O/r04_train_gemm_shapes.txt 2>&1; head -45 $O/r04_train_gemm_shapes.txt | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt; timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o tr -- python3 $R/bench.py --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r04_prof_train_trace.log 2>&1
python3 $R/tools/step_timeline.py /tmp/pt --bin-ms 5 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r04_train_step_timeline.txt 2>&1; head -60 $O/r04_train_step_timeline.txt
