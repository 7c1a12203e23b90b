#!/bin/bash
# Round-3 measured artefacts, part 3: the configs[3] stream after the row-chain / token-row / down-sampler work, and the driver's headline command on the same tree.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
T="timeout -k 10"
export RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/ps; $T 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o stream -- $B --mode sam2_stream --steps 5 --warmup 2 > $O/r03_prof_stream.log 2>&1
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $O/r03_bench_sam2_stream_kernel_stats.csv
python3 $R/tools/frame_timeline.py /tmp/ps --anchor "conv3x3s2_ln_gelu_kernel<1" --group-ms 0.3 > $O/r03_stream_frame_timeline.txt 2>&1
python3 $R/tools/kernel_stats_summary.py /tmp/ps memattn_cross $O/r03_bench_sam2_stream_memattn_summary.json
rm -rf /tmp/sf /tmp/sw
$T 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sf -- $B --mode sam2_stream --steps 1 --warmup 1 --no-graph > $O/r03_pmc_stream_fetch.log 2>&1
$T 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/sw -- $B --mode sam2_stream --steps 1 --warmup 1 --no-graph > $O/r03_pmc_stream_write.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/sf /tmp/sw memattn_cross_kernel $O/r03_bench_sam2_stream_memattn_traffic.json
unset RGA3_BENCH_TIMED_ONLY
mkdir -p $R/profiles; cp $O/r03_bench_sam2_stream_memattn_traffic.json $R/profiles/ 2>/dev/null     # the line below reads it
$T 900 $B --mode sam2_stream --steps 5 --warmup 2 > $O/r03_bench_sam2_stream.json 2> $O/r03_bench_sam2_stream.err; tail -c 400 $O/r03_bench_sam2_stream.json
$T 300 python3 $R/tools/evaluate_probe.py > $O/r03_evaluate_probe.log 2>&1; tail -1 $O/r03_evaluate_probe.log
$T 300 python3 $R/tools/memlayer_probe.py > $O/r03_memlayer_probe.log 2>&1; tail -4 $O/r03_memlayer_probe.log
# the driver's command
$T 1500 $B --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_headline_final.json 2> $O/r03_bench_headline_final.err; tail -c 300 $O/r03_bench_headline_final.json
