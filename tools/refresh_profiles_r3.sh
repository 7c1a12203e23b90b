#!/bin/bash
# Round-3 measured artefacts, one GPU-box call:  /usr/local/graft/bin/gpurun --timeout 3400 -- 'bash tools/refresh_profiles_r3.sh'
# rocprofv3 runs from /tmp with TMPDIR=/tmp, the program directly after "--", PMC passes separate from each other and from --stats (no trace domains beside --pmc
# (every profiler call runs under `timeout`: a --pmc pass that aborts inside rocprofv3 otherwise sits on the box until the call limit -- it happened; part 2 = tools/refresh_profiles_r3b.sh)
# except --kernel-trace).  Results land in gpurun_out/r03_* ; what is to be judged is copied into profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# 1. the driver's command (default mode = BASELINE metric, fresh batch per step) and its kernel statistics
$B --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_headline.json 2> $O/r03_bench_headline.err; tail -c 300 $O/r03_bench_headline.json
rm -rf /tmp/p1; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o head -- $B --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_prof_headline.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/r03_bench_headline_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p1 gemm_ $O/r03_bench_headline_gemm_summary.json
# 2. configs[1] forward alone: tuner decisions saved, then statistics and PMC passes of the TIMED tilings only
RGA3_TUNE_SAVE=$O/r03_tuner_forward.json $B --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_forward.json 2> $O/r03_bench_forward.err; tail -c 300 $O/r03_bench_forward.json
export RGA3_TUNE_LOAD=$O/r03_tuner_forward.json RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/p2; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o fwd -- $B --mode forward --steps 20 --warmup 3 --no-refine --no-cpu-baseline > $O/r03_prof_forward.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/r03_bench_forward_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p2 gemm_nt_ $O/r03_bench_forward_gemm_summary.json
rm -rf /tmp/pf /tmp/pw
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r03_pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r03_pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw gemm_nt_ $O/r03_bench_forward_gemm_traffic.json
unset RGA3_TUNE_LOAD RGA3_BENCH_TIMED_ONLY
# 3. the training step alone: PMC passes
rm -rf /tmp/tf /tmp/tw
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tf -- $B --mode train_full --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_pmc_train_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/tw -- $B --mode train_full --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_pmc_train_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/tf /tmp/tw gemm_ $O/r03_bench_train_full_gemm_traffic.json
# 4. configs[3] stream: kernel statistics + PMC traffic of the memory cross-attention kernel, then the line itself (cpu_baseline + dominant-kernel roofline live)
export RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/ps; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o stream -- $B --mode sam2_stream --steps 5 --warmup 2 > $O/r03_prof_stream.log 2>&1
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $O/r03_bench_sam2_stream_kernel_stats.csv
python3 $R/tools/frame_timeline.py /tmp/ps --anchor "conv3x3s2_kernel<true>" > $O/r03_stream_frame_timeline.txt 2>&1
python3 $R/tools/kernel_stats_summary.py /tmp/ps memattn_cross $O/r03_bench_sam2_stream_memattn_summary.json
rm -rf /tmp/sf /tmp/sw
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sf -- $B --mode sam2_stream --steps 2 --warmup 1 > $O/r03_pmc_stream_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/sw -- $B --mode sam2_stream --steps 2 --warmup 1 > $O/r03_pmc_stream_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/sf /tmp/sw memattn_cross_kernel $O/r03_bench_sam2_stream_memattn_traffic.json
unset RGA3_BENCH_TIMED_ONLY
mkdir -p $R/profiles; cp $O/r03_bench_sam2_stream_memattn_traffic.json $R/profiles/ 2>/dev/null     # the line below reads it
$B --mode sam2_stream --steps 5 --warmup 2 > $O/r03_bench_sam2_stream.json 2> $O/r03_bench_sam2_stream.err; tail -c 300 $O/r03_bench_sam2_stream.json
# 5. configs[4] LoRA fp8 step: kernel statistics, PMC traffic of its GEMM family, then the line (cpu_baseline live)
rm -rf /tmp/p8; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -o f8 -- $B --mode lora_fp8 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_prof_lora_fp8.log 2>&1
cp $(find /tmp/p8 -name "*kernel_stats.csv" | head -1) $O/r03_bench_lora_fp8_kernel_stats.csv
rm -rf /tmp/f8f /tmp/f8w
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/f8f -- $B --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline > $O/r03_pmc_f8_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/f8w -- $B --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline > $O/r03_pmc_f8_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/f8f /tmp/f8w gemm_ $O/r03_bench_lora_fp8_gemm_traffic.json
cp $O/r03_bench_lora_fp8_gemm_traffic.json $R/profiles/ 2>/dev/null
$B --mode lora_fp8 --steps 3 --warmup 1 > $O/r03_bench_lora_fp8.json 2> $O/r03_bench_lora_fp8.err; tail -c 300 $O/r03_bench_lora_fp8.json
# 6. probes
python3 $R/tools/evaluate_probe.py > $O/r03_evaluate_probe.log 2>&1; tail -1 $O/r03_evaluate_probe.log
python3 $R/tools/generate_probe.py 64 > $O/r03_generate_probe.log 2>&1; tail -1 $O/r03_generate_probe.log
# 7. where the step goes: per-shape GEMM table, timeline + non-GEMM kernel table of one training step
python3 $R/tools/gemm_shape_table.py $O/r03_train_gemm_shapes.json > $O/r03_train_gemm_shapes.txt 2>&1; head -8 $O/r03_train_gemm_shapes.txt | grep -v amdgpu
rm -rf /tmp/pt; timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o tr -- $B --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r03_prof_train_trace.log 2>&1
python3 $R/tools/step_timeline.py /tmp/pt --bin-ms 5 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r03_train_step_timeline.txt 2>&1; tail -n +2 $O/r03_train_step_timeline.txt | head -3
# 8. the GPU test suite, the driver's command
cd $R && python3 -m pytest tests/ -x -q -m gpu > $O/r03_gpu_tests.log 2>&1; tail -2 $O/r03_gpu_tests.log
