#!/bin/bash
# Round-6 measured artefacts, one GPU-box call:  /usr/local/graft/bin/gpurun --timeout 3400 -- 'bash tools/refresh_profiles_r6.sh'
# rocprofv3 runs from /tmp with TMPDIR=/tmp, the program directly after "--", every profiler call under `timeout`, PMC passes separate from each other and from --stats.
# Results land in gpurun_out/r06_* ; what is to be judged is copied into profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# 0. matrix-pipe counters of the shipped kernels at their bench shapes (feeds roofline.mfma_busy of the lines below)
rm -rf /tmp/pu; i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d /tmp/pu/p$i -- python3 $R/tools/pmc_pipe_util.py run > $O/r06_pmc_pipe_p$i.log 2>&1; echo "pipe pass $i rc $?"
done
python3 $R/tools/pmc_pipe_util.py sum /tmp/pu $O/r06_pmc_pipe_util.json > $O/r06_pmc_pipe_util.txt 2>&1; mkdir -p $R/profiles; cp $O/r06_pmc_pipe_util.json $R/profiles/
# 0b. Hiera stage-3 products (K = 576 family): time per tiling + PMC passes (L2 hit rate, fetch / write bytes, MFMA-busy, VALU share)
timeout -k 10 400 python3 $R/tools/probes/k576_probe.py time $O/r06_k576_time.json > $O/r06_k576_time.log 2>&1; grep -v amdgpu $O/r06_k576_time.log | tail -4
rm -rf /tmp/pk; i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pk/p$i -- python3 $R/tools/probes/k576_probe.py pmc > $O/r06_k576_pmc_p$i.log 2>&1; echo "k576 pmc pass $i rc $?"
done
python3 $R/tools/probes/k576_probe.py sum /tmp/pk $O/r06_k576_pmc.json > $O/r06_k576_pmc.txt 2>&1
# 1. the driver's command (default mode = BASELINE metric) and its kernel statistics
$B --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_headline.json 2> $O/r06_bench_headline.err; tail -c 300 $O/r06_bench_headline.json
rm -rf /tmp/p1; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o head -- $B --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_prof_headline.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/r06_bench_headline_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p1 gemm_ $O/r06_bench_headline_gemm_summary.json
# 2. configs[1] forward alone: tuner decisions saved, then statistics and PMC passes of the TIMED tilings only
RGA3_TUNE_SAVE=$O/r06_tuner_forward.json $B --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_bench_forward.json 2> $O/r06_bench_forward.err; tail -c 300 $O/r06_bench_forward.json
export RGA3_TUNE_LOAD=$O/r06_tuner_forward.json RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/p2; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o fwd -- $B --mode forward --steps 20 --warmup 3 --no-refine --no-cpu-baseline > $O/r06_prof_forward.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/r06_bench_forward_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/p2 gemm_nt_ $O/r06_bench_forward_gemm_summary.json
rm -rf /tmp/pf /tmp/pw
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r06_pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $B --mode forward --steps 3 --warmup 1 --no-refine --no-cpu-baseline > $O/r06_pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw gemm_nt_ $O/r06_bench_forward_gemm_traffic.json
unset RGA3_TUNE_LOAD RGA3_BENCH_TIMED_ONLY
mkdir -p $R/profiles; cp $O/r06_bench_forward_gemm_traffic.json $O/r06_pmc_pipe_util.json $R/profiles/ 2>/dev/null     # the lines below read them (same tree: accepted)
# 2b. PMC traffic of the other lines' dominant families, collected on THIS tree (one repeated batch: the same products every step)
T="timeout -k 10 420"
rm -rf /tmp/tf /tmp/tw
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tf -- $B --mode train_full --steps 2 --warmup 1 --no-cpu-baseline --batches repeat > $O/r06_pmc_train_fetch.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/tw -- $B --mode train_full --steps 2 --warmup 1 --no-cpu-baseline --batches repeat > $O/r06_pmc_train_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/tf /tmp/tw gemm_ $O/r06_bench_train_full_gemm_traffic.json
rm -rf /tmp/sf /tmp/sw
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sf -- $B --mode sam2_stream --steps 1 --warmup 1 --no-graph --no-cpu-baseline > $O/r06_pmc_stream_fetch.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/sw -- $B --mode sam2_stream --steps 1 --warmup 1 --no-graph --no-cpu-baseline > $O/r06_pmc_stream_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/sf /tmp/sw memattn_cross_kernel $O/r06_bench_sam2_stream_memattn_traffic.json
rm -rf /tmp/f8f /tmp/f8w
# configs[4] / lora_fp8 under rocprofv3 --pmc died in every attempt of round 5 (32 frames x 4 accumulation steps of counter records): ONE micro-step, GEMM kernels only
for a in 1 2; do rm -rf /tmp/f8f; $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "gemm_fp8|gemm_nt" --output-format csv -d /tmp/f8f -- $B --mode lora_fp8 --grad-accum 1 --steps 1 --warmup 0 --no-cpu-baseline --batches repeat > $O/r06_pmc_f8_fetch.log 2>&1 && [ -n "$(find /tmp/f8f -name '*counter_collection.csv' | head -1)" ] && break; echo "f8 fetch pass attempt $a failed"; done
for a in 1 2; do rm -rf /tmp/f8w; $T rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "gemm_fp8|gemm_nt" --output-format csv -d /tmp/f8w -- $B --mode lora_fp8 --grad-accum 1 --steps 1 --warmup 0 --no-cpu-baseline --batches repeat > $O/r06_pmc_f8_write.log 2>&1 && [ -n "$(find /tmp/f8w -name '*counter_collection.csv' | head -1)" ] && break; echo "f8 write pass attempt $a failed"; done
python3 $R/tools/pmc_traffic.py /tmp/f8f /tmp/f8w gemm_ $O/r06_bench_lora_fp8_gemm_traffic.json; cp $O/r06_bench_lora_fp8_gemm_traffic.json $R/profiles/ 2>/dev/null
cp $O/r06_bench_train_full_gemm_traffic.json $O/r06_bench_sam2_stream_memattn_traffic.json $R/profiles/ 2>/dev/null
# 2c. the driver's command once more, now that every counter profile of this tree exists (its line then carries traffic / mfma_busy of THIS tree)
$B --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_headline.json 2> $O/r06_bench_headline.err; tail -c 300 $O/r06_bench_headline.json
$B --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r06_bench_forward.json 2> $O/r06_bench_forward.err; tail -c 300 $O/r06_bench_forward.json
# 3. configs[3] stream (1 and 4 objects) and configs[4] fp8 lines
$B --mode sam2_stream --objects 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_sam2_stream_4obj.json 2> $O/r06_bench_sam2_stream_4obj.err; tail -c 300 $O/r06_bench_sam2_stream_4obj.json
$B --mode sam2_stream --steps 5 --warmup 2 > $O/r06_bench_sam2_stream.json 2> $O/r06_bench_sam2_stream.err; tail -c 300 $O/r06_bench_sam2_stream.json
$B --mode lora_fp8 --steps 3 --warmup 1 > $O/r06_bench_lora_fp8.json 2> $O/r06_bench_lora_fp8.err; tail -c 300 $O/r06_bench_lora_fp8.json
# 4. where the step goes: per-shape GEMM table, timeline + non-GEMM kernel table of one training step
python3 $R/tools/gemm_shape_table.py $O/r06_train_gemm_shapes.json > $O/r06_train_gemm_shapes.txt 2>&1; head -8 $O/r06_train_gemm_shapes.txt | grep -v amdgpu
rm -rf /tmp/pt; timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o tr -- $B --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r06_prof_train_trace.log 2>&1
python3 $R/tools/step_timeline.py /tmp/pt --bin-ms 5 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r06_train_step_timeline.txt 2>&1; tail -n +2 $O/r06_train_step_timeline.txt | head -3
# 5. probes + the GPU test suite, the driver's command
timeout -k 10 600 python3 $R/tools/blas_reference_point.py > $O/r06_blas.log 2>&1; cp $O/blas_reference_point.json $O/r06_blas_reference_point.json
(cd $R && timeout -k 10 600 python3 tools/hiera_attn_probe.py > $O/r06_hiera_attn_probe.log 2>&1; grep "windows of 256" $O/r06_hiera_attn_probe.log | cut -c1-200)
(cd $R && NF=16 AB=_LN_SUMS timeout -k 10 600 python3 tools/sam2_encoder_probe.py 5 > $O/r06_encoder_probe.log 2>&1; grep "ms per\|A/B" $O/r06_encoder_probe.log)
python3 $R/tools/evaluate_probe.py > $O/r06_evaluate_probe.log 2>&1; tail -1 $O/r06_evaluate_probe.log
python3 $R/tools/generate_probe.py 64 > $O/r06_generate_probe.log 2>&1; tail -1 $O/r06_generate_probe.log
cd $R && python3 -m pytest tests/ -x -q -m gpu > $O/r06_gpu_tests.log 2>&1; tail -2 $O/r06_gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; tail -3 $O/r06_smoke.log
# 6. one steady-state stream frame launch by launch, one configs[4] optimizer step and the encoder alone by kernel, the step's framework kernels
cd $R && FRAME=93 bash tools/gpu_r6.sh frameline > /dev/null 2>&1; head -3 $O/r06_stream_frame_timeline.txt
bash tools/gpu_r6.sh fp8prof encprof atencensus > $O/r06_extra_profiles.log 2>&1; grep "ms per" $O/r06_extra_profiles.log
timeout -k 10 300 python3 tools/probes/swiglu_quant_probe.py > $O/r06_swiglu_quant_probe.log 2>&1
