#!/bin/bash
# round-2 GPU pass A: full GPU test-suite, smoke, the default bench line (driver's flags)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2a_tests.log
( time timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' ) > gpurun_out/r2a_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r2a_smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc=$?" >> gpurun_out/r2a_bench.err
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r2a_bench_gpus2.out 2>&1
echo "gpus2 rc=$?" >> gpurun_out/r2a_bench_gpus2.out
tail -3 gpurun_out/r2a_tests.log; tail -3 gpurun_out/r2a_smoke.log; cat gpurun_out/r2a_bench.json; tail -5 gpurun_out/r2a_bench.err; tail -2 gpurun_out/r2a_bench_gpus2.out
