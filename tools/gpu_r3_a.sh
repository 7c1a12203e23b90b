#!/bin/bash
# round 3, job A: the two full-size backward parity tests, then a kernel trace of the configs[3] stream (timed region only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && python3 tools/decoder_fullsize_grad.py > $O/r03_decoder_grad.log 2>&1
cd /tmp && export TMPDIR=/tmp
export RGA3_BENCH_TIMED_ONLY=1
rm -rf $O/prof_stream; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o stream -- python3 $R/bench.py --mode sam2_stream --steps 3 --warmup 2 > $O/prof_stream.log 2>&1
python3 $R/tools/frame_timeline.py $O/prof_stream --list > $O/r03_stream_frame_timeline_base.txt 2>&1
find $O/prof_stream -name "*kernel_trace.csv" -size +30M -delete
cat $O/r03_decoder_grad.log | tail -120; head -50 $O/r03_stream_frame_timeline_base.txt | cut -c1-160
