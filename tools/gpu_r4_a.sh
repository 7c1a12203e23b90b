#!/bin/bash
# round 4, job A: baseline of the inherited tree -- GPU tests, vendor-BLAS reference point, matrix-pipe counters of the shipped kernels, forward bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R && timeout 900 python3 -m pytest tests/ -x -q -m gpu > $O/r04a_gpu_tests.log 2>&1; tail -2 $O/r04a_gpu_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/tools/blas_reference_point.py > $O/r04a_blas.log 2>&1; cp $O/blas_reference_point.json $O/r04a_blas_reference_point.json; tail -12 $O/r04a_blas.log
rm -rf /tmp/pu
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d /tmp/pu/p$i -- python3 $R/tools/pmc_pipe_util.py run > $O/r04a_pmc_p$i.log 2>&1
  echo "pass $i ($set) rc $?"
done
python3 $R/tools/pmc_pipe_util.py sum /tmp/pu $O/r04a_pmc_pipe_util.json 2>&1 | tail -80
python3 $R/bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r04a_bench_forward.json 2> $O/r04a_bench_forward.err; tail -c 1500 $O/r04a_bench_forward.json
