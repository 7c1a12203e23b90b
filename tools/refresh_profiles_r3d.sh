#!/bin/bash
# Round-3 measured artefacts, part 4: the headline's and the forward's kernel statistics + PMC traffic once more on the final tree (every profiler call under `timeout`).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
T="timeout -k 10"
rm -rf /tmp/p1; $T 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o head -- $B --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_prof_headline.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/r03_bench_headline_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p1 gemm_ $O/r03_bench_headline_gemm_summary.json
RGA3_TUNE_SAVE=$O/r03_tuner_forward.json $T 900 $B --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_forward.json 2> $O/r03_bench_forward.err; tail -c 200 $O/r03_bench_forward.json
export RGA3_TUNE_LOAD=$O/r03_tuner_forward.json RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/p2; $T 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o fwd -- $B --mode forward --steps 20 --warmup 3 --no-refine --no-cpu-baseline > $O/r03_prof_forward.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/r03_bench_forward_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p2 gemm_nt_ $O/r03_bench_forward_gemm_summary.json
rm -rf /tmp/pf /tmp/pw
$T 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r03_pmc_fetch.log 2>&1
$T 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r03_pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw gemm_nt_ $O/r03_bench_forward_gemm_traffic.json; cat $O/r03_bench_forward_gemm_traffic.json | tr -d '\n' | cut -c1-300
