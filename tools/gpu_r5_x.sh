#!/bin/bash
# final check of the committed tree: the driver's two commands + the fp8 line (its earlier copy quoted a counter file without launches)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 2400 python3 -m pytest tests/ -x -q -m gpu > $O/r05_gpu_tests.log 2>&1; echo "suite rc $?"; tail -2 $O/r05_gpu_tests.log | cut -c1-300
timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/r05_smoke.log
python3 bench.py --mode lora_fp8 --steps 3 --warmup 1 > $O/r05_bench_lora_fp8.json 2> $O/r05_bench_lora_fp8.err; tail -c 300 $O/r05_bench_lora_fp8.json
python3 bench.py --mode sam2_stream --objects 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/r05_bench_sam2_stream_2obj.json 2> $O/r05_bench_sam2_stream_2obj.err; tail -c 200 $O/r05_bench_sam2_stream_2obj.json
python3 bench.py --gpus 1 > $O/r05_bench_headline_default_flags.json 2> $O/r05_bench_headline_default_flags.err; tail -c 200 $O/r05_bench_headline_default_flags.json
