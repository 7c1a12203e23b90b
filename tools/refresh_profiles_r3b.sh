#!/bin/bash
# Round-3 measured artefacts, part 2 (every profiler call under its own `timeout`: a PMC pass that aborts must not sit on the box).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
T="timeout -k 10"
# 4. configs[3] stream: kernel statistics, one-frame timeline, PMC traffic of the memory cross-attention kernel (eager launches: counters and graph replay do not mix), the line
export RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/ps; $T 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o stream -- $B --mode sam2_stream --steps 5 --warmup 2 > $O/r03_prof_stream.log 2>&1
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $O/r03_bench_sam2_stream_kernel_stats.csv
python3 $R/tools/frame_timeline.py /tmp/ps --anchor "conv3x3s2_kernel<true>" > $O/r03_stream_frame_timeline.txt 2>&1
python3 $R/tools/kernel_stats_summary.py /tmp/ps memattn_cross $O/r03_bench_sam2_stream_memattn_summary.json
rm -rf /tmp/sf /tmp/sw
$T 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sf -- $B --mode sam2_stream --steps 1 --warmup 1 --no-graph > $O/r03_pmc_stream_fetch.log 2>&1
$T 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/sw -- $B --mode sam2_stream --steps 1 --warmup 1 --no-graph > $O/r03_pmc_stream_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/sf /tmp/sw memattn_cross_kernel $O/r03_bench_sam2_stream_memattn_traffic.json
unset RGA3_BENCH_TIMED_ONLY
mkdir -p $R/profiles; cp $O/r03_bench_sam2_stream_memattn_traffic.json $R/profiles/ 2>/dev/null     # the line below reads it
$T 900 $B --mode sam2_stream --steps 5 --warmup 2 > $O/r03_bench_sam2_stream.json 2> $O/r03_bench_sam2_stream.err; tail -c 300 $O/r03_bench_sam2_stream.json
# 5. configs[4] LoRA fp8 step: kernel statistics, PMC traffic of its GEMM family (one repeated batch: the same products), the line
rm -rf /tmp/p8; $T 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -o f8 -- $B --mode lora_fp8 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_prof_lora_fp8.log 2>&1
cp $(find /tmp/p8 -name "*kernel_stats.csv" | head -1) $O/r03_bench_lora_fp8_kernel_stats.csv
rm -rf /tmp/f8f /tmp/f8w
$T 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/f8f -- $B --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline --batches repeat > $O/r03_pmc_f8_fetch.log 2>&1
$T 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/f8w -- $B --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline --batches repeat > $O/r03_pmc_f8_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/f8f /tmp/f8w gemm_ $O/r03_bench_lora_fp8_gemm_traffic.json
cp $O/r03_bench_lora_fp8_gemm_traffic.json $R/profiles/ 2>/dev/null
$T 1200 $B --mode lora_fp8 --steps 3 --warmup 1 > $O/r03_bench_lora_fp8.json 2> $O/r03_bench_lora_fp8.err; tail -c 300 $O/r03_bench_lora_fp8.json
# 3'. the training step's GEMM traffic (one repeated batch)
rm -rf /tmp/tf /tmp/tw
$T 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tf -- $B --mode train_full --steps 2 --warmup 1 --no-cpu-baseline --batches repeat > $O/r03_pmc_train_fetch.log 2>&1
$T 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/tw -- $B --mode train_full --steps 2 --warmup 1 --no-cpu-baseline --batches repeat > $O/r03_pmc_train_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/tf /tmp/tw gemm_ $O/r03_bench_train_full_gemm_traffic.json
# 7. where the step goes: timeline + non-GEMM kernel table of one training step, per-shape GEMM table
rm -rf /tmp/pt; $T 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o tr -- $B --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r03_prof_train_trace.log 2>&1
python3 $R/tools/step_timeline.py /tmp/pt --bin-ms 5 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r03_train_step_timeline.txt 2>&1; tail -n +2 $O/r03_train_step_timeline.txt | head -3
$T 600 python3 $R/tools/gemm_shape_table.py $O/r03_train_gemm_shapes.json > $O/r03_train_gemm_shapes.txt 2>&1; head -8 $O/r03_train_gemm_shapes.txt | grep -v amdgpu
# 6. probes
$T 300 python3 $R/tools/evaluate_probe.py > $O/r03_evaluate_probe.log 2>&1; tail -1 $O/r03_evaluate_probe.log
$T 300 python3 $R/tools/generate_probe.py 64 > $O/r03_generate_probe.log 2>&1; tail -1 $O/r03_generate_probe.log
