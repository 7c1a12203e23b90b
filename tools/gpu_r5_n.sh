#!/bin/bash
# Early round-5 evidence of the tree as it stands (sections 1 and 2 of tools/refresh_profiles_r5.sh): headline + forward lines, kernel statistics, forward GEMM traffic.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
$B --gpus 1 --steps 20 --warmup 5 > $O/r05n_bench_headline.json 2> $O/r05n_bench_headline.err; tail -c 300 $O/r05n_bench_headline.json
rm -rf /tmp/p1; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o head -- $B --steps 20 --warmup 5 --no-cpu-baseline > $O/r05n_prof_headline.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/r05n_bench_headline_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p1 gemm_ $O/r05n_bench_headline_gemm_summary.json
RGA3_TUNE_SAVE=$O/r05n_tuner_forward.json $B --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r05n_bench_forward.json 2> $O/r05n_bench_forward.err; tail -c 300 $O/r05n_bench_forward.json
export RGA3_TUNE_LOAD=$O/r05n_tuner_forward.json RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/p2; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o fwd -- $B --mode forward --steps 20 --warmup 3 --no-refine --no-cpu-baseline > $O/r05n_prof_forward.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/r05n_bench_forward_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p2 gemm_nt_ $O/r05n_bench_forward_gemm_summary.json
rm -rf /tmp/pf /tmp/pw
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r05n_pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r05n_pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw gemm_nt_ $O/r05n_bench_forward_gemm_traffic.json
unset RGA3_TUNE_LOAD RGA3_BENCH_TIMED_ONLY
rm -rf /tmp/pt; timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o tr -- $B --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r05n_prof_train_trace.log 2>&1
python3 $R/tools/step_timeline.py /tmp/pt --bin-ms 5 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r05n_train_step_timeline.txt 2>&1; tail -n +2 $O/r05n_train_step_timeline.txt | head -3
