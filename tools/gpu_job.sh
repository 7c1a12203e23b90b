#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
RGA3_TUNE_SAVE=$O/r2l_tuner_forward.json $B --mode forward --steps 10 --warmup 3 --no-cpu-baseline > $O/r2l_fwd.json 2> $O/r2l_fwd.err
export RGA3_TUNE_LOAD=$O/r2l_tuner_forward.json RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o fwd -- $B --mode forward --steps 20 --warmup 3 --no-refine --no-cpu-baseline > $O/r2l_prof_forward.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/r2l_forward_kernel_stats.csv
head -30 $O/r2l_forward_kernel_stats.csv | cut -c1-230
