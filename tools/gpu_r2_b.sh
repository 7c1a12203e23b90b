#!/bin/bash
# round-2 GPU pass B: the default bench line (driver's flags) + rocprofv3 kernel trace of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
echo "bench rc=$?" >> gpurun_out/r2b_bench.err
cat gpurun_out/r2b_bench.json; tail -5 gpurun_out/r2b_bench.err
rm -rf /tmp/prof_b; mkdir -p /tmp/prof_b
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o r2b -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/r2b_prof.log 2>&1
echo "prof rc=$?" >> gpurun_out/r2b_prof.log
find /tmp/prof_b -name "*kernel_stats*" -exec cp {} gpurun_out/r2b_kernel_stats.csv \;
tail -3 gpurun_out/r2b_prof.log; head -25 gpurun_out/r2b_kernel_stats.csv | cut -c1-200
