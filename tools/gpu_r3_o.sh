#!/bin/bash
# configs[4] (lora_fp8): PMC traffic passes once more, with everything that has tripped the profiler switched off (fresh pinned uploads, the side-stream vision prefetch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
B="python3 $R/bench.py"
export RGA3_BENCH_TIMED_ONLY=1
for try in 1 2; do
rm -rf /tmp/f8f
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/f8f -- $B --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline --batches repeat --no-prefetch > $O/r03_pmc_f8_fetch2.log 2>&1; rc=$?; echo "fetch try $try rc $rc"
[ $rc -eq 0 ] && break
done
rm -rf /tmp/f8w
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/f8w -- $B --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline --batches repeat --no-prefetch > $O/r03_pmc_f8_write2.log 2>&1; echo "write rc $?"
python3 $R/tools/pmc_traffic.py /tmp/f8f /tmp/f8w gemm_ $O/r03_bench_lora_fp8_gemm_traffic.json; cat $O/r03_bench_lora_fp8_gemm_traffic.json | tr -d '\n' | cut -c1-400
