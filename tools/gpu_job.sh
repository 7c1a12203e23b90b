#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt; RGA3_TUNE_LOAD=$R/profiles/r02_tuner_forward.json timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o tr -- python3 $R/bench.py --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r2r_trace.log 2>&1
python3 $R/tools/trace_gaps.py /tmp/tt --last-ms 1500 > $O/r2r_gaps.txt 2>&1; head -45 $O/r2r_gaps.txt | cut -c1-180
