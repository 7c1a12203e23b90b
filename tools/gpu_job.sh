#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/grad_locate.py > gpurun_out/r2f_locate.log 2>&1
tail -14 gpurun_out/r2f_locate.log | cut -c1-200
rm -rf /tmp/tr; ( cd /tmp && RGA3_BENCH_TIMED_ONLY=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o fwd -- python3 "$GRAFT_REPO_ROOT/bench.py" --mode forward --steps 6 --warmup 3 --no-cpu-baseline ) > gpurun_out/r2f_trace.log 2>&1
python3 tools/trace_gaps.py /tmp/tr --last-ms 130 > gpurun_out/r2f_gaps.txt 2>&1; head -40 gpurun_out/r2f_gaps.txt | cut -c1-200
