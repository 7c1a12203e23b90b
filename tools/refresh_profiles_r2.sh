#!/bin/bash
# Round-2 measured artefacts, one GPU-box call:  /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/refresh_profiles_r2.sh'
# rocprofv3 runs from /tmp with TMPDIR=/tmp, the program directly after "--", PMC passes separate from each other and from --stats.
# Results land in gpurun_out/r02_* ; copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# 1. the driver's command (default mode = BASELINE metric) and its kernel statistics
$B --gpus 1 --steps 20 --warmup 5 > $O/r02_bench_headline.json 2> $O/r02_bench_headline.err; tail -c 400 $O/r02_bench_headline.json
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o head -- $B --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_prof_headline.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/r02_bench_headline_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p1 gemm_ $O/r02_bench_headline_gemm_summary.json
# 2. configs[1] forward alone: tuner decisions saved, then statistics and PMC passes of the TIMED tilings only (no tuner trials, no verification launches)
RGA3_TUNE_SAVE=$O/r02_tuner_forward.json $B --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_bench_forward.json 2> $O/r02_bench_forward.err; tail -c 300 $O/r02_bench_forward.json
export RGA3_TUNE_LOAD=$O/r02_tuner_forward.json RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o fwd -- $B --mode forward --steps 20 --warmup 3 --no-refine --no-cpu-baseline > $O/r02_prof_forward.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/r02_bench_forward_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p2 gemm_nt_ $O/r02_bench_forward_gemm_summary.json
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r02_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r02_pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw gemm_nt_ $O/r02_bench_forward_gemm_traffic.json
unset RGA3_TUNE_LOAD RGA3_BENCH_TIMED_ONLY
# 3. the training step alone: decisions, then PMC passes
RGA3_TUNE_SAVE=$O/r02_tuner_train.json $B --mode train_full --steps 10 --warmup 3 --no-cpu-baseline > $O/r02_bench_train_full.json 2> $O/r02_bench_train_full.err; tail -c 300 $O/r02_bench_train_full.json
export RGA3_TUNE_LOAD=$O/r02_tuner_train.json
rm -rf /tmp/tf /tmp/tw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tf -- $B --mode train_full --steps 2 --warmup 1 --no-cpu-baseline > $O/r02_pmc_train_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/tw -- $B --mode train_full --steps 2 --warmup 1 --no-cpu-baseline > $O/r02_pmc_train_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/tf /tmp/tw gemm_ $O/r02_bench_train_full_gemm_traffic.json
unset RGA3_TUNE_LOAD
# 4. the other configs and probes
$B --mode sam2_stream --steps 3 --warmup 1 > $O/r02_bench_sam2_stream.json 2> $O/r02_bench_sam2_stream.err; tail -c 300 $O/r02_bench_sam2_stream.json
$B --mode lora_fp8 --steps 3 --warmup 1 > $O/r02_bench_lora_fp8.json 2> $O/r02_bench_lora_fp8.err; tail -c 300 $O/r02_bench_lora_fp8.json
python3 $R/tools/mall_probe.py $O/r02_mall_probe.json > $O/r02_mall_probe.log 2>&1; tail -9 $O/r02_mall_probe.log
python3 $R/tools/evaluate_probe.py > $O/r02_evaluate_probe.log 2>&1; tail -1 $O/r02_evaluate_probe.log
python3 $R/tools/generate_probe.py 64 > $O/r02_generate_probe.log 2>&1; tail -1 $O/r02_generate_probe.log
# 5. functional two-rank run of the training step on the one GPU (gloo, both ranks on device 0): bucket order, sparse row exchange, side streams
RGA3_BENCH_SHARE_GPU=1 RGA3_BENCH_BACKEND=gloo timeout 900 $B --gpus 2 --mode train_full --steps 2 --warmup 1 --no-cpu-baseline > $O/r02_two_ranks_shared_gpu.json 2> $O/r02_two_ranks_shared_gpu.err; tail -c 600 $O/r02_two_ranks_shared_gpu.json; tail -3 $O/r02_two_ranks_shared_gpu.err
# 6. where the step goes: per-shape GEMM table, timeline + non-GEMM kernel table of one training step, gaps
python3 $R/tools/gemm_shape_table.py $O/r02_train_gemm_shapes.json > $O/r02_train_gemm_shapes.txt 2>&1; head -12 $O/r02_train_gemm_shapes.txt | grep -v amdgpu
rm -rf /tmp/pt; rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o tr -- $B --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r02_prof_train_trace.log 2>&1
python3 $R/tools/step_timeline.py /tmp/pt --bin-ms 5 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r02_train_step_timeline.txt 2>&1; tail -n +2 $O/r02_train_step_timeline.txt | head -3
python3 $R/tools/trace_gaps.py /tmp/pt --last-ms 600 > $O/r02_train_gaps.txt 2>&1; head -1 $O/r02_train_gaps.txt
# 7. the GPU test suite
cd $R && python3 -m pytest tests -m gpu -q > $O/r02_gpu_tests.log 2>&1; tail -2 $O/r02_gpu_tests.log
