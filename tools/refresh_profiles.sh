#!/bin/bash
# Regenerates every measured artefact under gpurun_out/ in one GPU-box call (copy the results into profiles/ afterwards):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh'
# rocprofv3 runs from /tmp with TMPDIR=/tmp, the program directly after "--", PMC passes separate from each other and from --stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 -m pytest $R/tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
RGA3_TUNE_DUMP=$O/tuner_table.json python3 $R/bench.py --steps 20 --warmup 3 > $O/bench_fwd.json 2> $O/bench_fwd.err; tail -c 600 $O/bench_fwd.json
rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o fwd -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_fwd_prof.log 2>&1
cp $(find /tmp/pf -name "*kernel_stats.csv" | head -1) $O/fwd_kernel_stats.csv
python3 $R/tools/kernel_stats_summary.py /tmp/pf gemm_nt_ $O/gemm_summary.json
rm -rf /tmp/pmf /tmp/pmw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-refine > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-refine > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pmf /tmp/pmw gemm_nt_ $O/gemm_traffic.json
python3 $R/bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; tail -c 300 $O/bench_train.json
python3 $R/bench.py --mode train_full --steps 10 --warmup 3 > $O/bench_train_full.json 2> $O/bench_train_full.err; tail -c 300 $O/bench_train_full.json
rm -rf /tmp/ptf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptf -o tf -- python3 $R/bench.py --mode train_full --steps 10 --warmup 3 > $O/bench_train_full_prof.log 2>&1
cp $(find /tmp/ptf -name "*kernel_stats.csv" | head -1) $O/train_full_kernel_stats.csv
python3 $R/bench.py --mode lora_fp8 --steps 3 --warmup 1 > $O/bench_lora_fp8.json 2> $O/bench_lora_fp8.err; tail -c 300 $O/bench_lora_fp8.json
python3 $R/bench.py --mode sam2_stream --steps 3 --warmup 1 > $O/bench_sam2_stream.json 2> $O/bench_sam2_stream.err; tail -c 300 $O/bench_sam2_stream.json
python3 $R/tools/bench_preproc.py > $O/bench_preproc.log 2>&1; tail -3 $O/bench_preproc.log
python3 $R/tools/blas_reference_point.py > $O/blas_reference_point.log 2>&1; tail -3 $O/blas_reference_point.log
python3 $R/tools/evaluate_probe.py > $O/evaluate_probe.log 2>&1; tail -1 $O/evaluate_probe.log
python3 $R/tools/generate_probe.py 64 > $O/generate_probe.log 2>&1; tail -1 $O/generate_probe.log
python3 $R/tools/sam2_encoder_probe.py 5 2>&1 | grep "ms per" > $O/sam2_encoder_probe.log; cat $O/sam2_encoder_probe.log
