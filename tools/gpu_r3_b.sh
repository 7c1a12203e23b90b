#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export RGA3_BENCH_TIMED_ONLY=1
rm -rf $O/prof_stream; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o stream -- python3 $R/bench.py --mode sam2_stream --steps 3 --warmup 2 > $O/prof_stream.log 2>&1
python3 $R/tools/frame_timeline.py $O/prof_stream --list --anchor "conv3x3s2_kernel<true>" > $O/r03_stream_frame_timeline_base.txt 2>&1
find $O/prof_stream -name "*kernel_trace.csv" -size +30M -delete
head -3 $O/r03_stream_frame_timeline_base.txt
