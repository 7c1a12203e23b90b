#!/bin/bash
# scratch job for /usr/local/graft/bin/gpurun -- 'bash tools/gpu_job.sh' (edited per experiment; the last useful content: the round's GPU suite + default bench)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/job_tests.log 2>&1; tail -3 $O/job_tests.log | cut -c1-200
python3 bench.py > $O/job_bench.json 2> $O/job_bench.err; tail -c 400 $O/job_bench.json
