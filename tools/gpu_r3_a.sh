#!/bin/bash
# round 3, job A: the two full-size backward parity tests, then a kernel trace of the configs[3] stream (timed region only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && python3 -m pytest tests/test_fullsize_parity_gpu.py -x -q -m gpu -k "lora_r128 or mask_decoder_sam2_l" 2>&1 | tail -25 > $O/r03_fullsize_bwd.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --mode sam2_stream --steps 5 --warmup 2 > $O/r03_stream_base.json 2> $O/r03_stream_base.err
export RGA3_BENCH_TIMED_ONLY=1
rm -rf $O/prof_stream; rocprofv3 --kernel-trace --stats -d $O/prof_stream -o stream -- python3 $R/bench.py --mode sam2_stream --steps 3 --warmup 2 > $O/prof_stream.log 2>&1
python3 $R/tools/frame_timeline.py $O/prof_stream --list > $O/r03_stream_frame_timeline_base.txt 2>&1
find $O/prof_stream -name "*kernel_trace.csv" -size +30M -delete
tail -25 $O/r03_fullsize_bwd.log; cat $O/r03_stream_base.json | cut -c1-400; head -50 $O/r03_stream_frame_timeline_base.txt | cut -c1-160
