#!/bin/bash
# completes the round-5 set: matrix-pipe counters (the summary step had lost its module path), the fp8 line's PMC traffic (profiler died in that mode: retried),
# then the bench lines once more so that they carry mfma_busy / traffic of this tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
rm -rf /tmp/pu; i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d /tmp/pu/p$i -- python3 $R/tools/pmc_pipe_util.py run > $O/r05_pmc_pipe_p$i.log 2>&1; echo "pipe pass $i rc $?"
done
python3 $R/tools/pmc_pipe_util.py sum /tmp/pu $O/r05_pmc_pipe_util.json > $O/r05_pmc_pipe_util.txt 2>&1; tail -3 $O/r05_pmc_pipe_util.txt | cut -c1-200; cp $O/r05_pmc_pipe_util.json $R/profiles/
T="timeout -k 10 420"
for a in 1 2 3; do rm -rf /tmp/f8f; $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/f8f -- $B --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline --batches repeat > $O/r05_pmc_f8_fetch.log 2>&1 && [ -n "$(find /tmp/f8f -name '*counter_collection.csv' | head -1)" ] && break; echo "f8 fetch pass attempt $a failed"; done
for a in 1 2 3; do rm -rf /tmp/f8w; $T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/f8w -- $B --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline --batches repeat > $O/r05_pmc_f8_write.log 2>&1 && [ -n "$(find /tmp/f8w -name '*counter_collection.csv' | head -1)" ] && break; echo "f8 write pass attempt $a failed"; done
python3 $R/tools/pmc_traffic.py /tmp/f8f /tmp/f8w gemm_ $O/r05_bench_lora_fp8_gemm_traffic.json; cp $O/r05_bench_lora_fp8_gemm_traffic.json $R/profiles/
$B --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_headline.json 2> $O/r05_bench_headline.err; tail -c 300 $O/r05_bench_headline.json
$B --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_bench_forward.json 2> $O/r05_bench_forward.err; tail -c 300 $O/r05_bench_forward.json
$B --mode sam2_stream --steps 5 --warmup 2 > $O/r05_bench_sam2_stream.json 2> $O/r05_bench_sam2_stream.err; tail -c 200 $O/r05_bench_sam2_stream.json
$B --mode lora_fp8 --steps 3 --warmup 1 > $O/r05_bench_lora_fp8.json 2> $O/r05_bench_lora_fp8.err; tail -c 300 $O/r05_bench_lora_fp8.json
