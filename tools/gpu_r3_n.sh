#!/bin/bash
# one-frame timeline of the configs[3] stream as it stands (kernel trace of the graph-replayed stream; anchor = first kernel of a frame's memory encoder)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export RGA3_BENCH_TIMED_ONLY=1
rm -rf /tmp/ps; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o stream -- python3 $R/bench.py --mode sam2_stream --steps 5 --warmup 2 > $O/r03_prof_stream2.log 2>&1
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $O/r03_bench_sam2_stream_kernel_stats_rowchain.csv
python3 $R/tools/frame_timeline.py /tmp/ps --anchor "conv3x3s2_ln_gelu_kernel<1" --group-ms 0.3 > $O/r03_stream_frame_timeline_rowchain.txt 2>&1
tail -3 $O/r03_prof_stream2.log
head -5 $O/r03_stream_frame_timeline_rowchain.txt
