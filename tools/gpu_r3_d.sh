#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_sam2_kernels_gpu.py tests/test_sam2_gpu.py tests/test_unigr_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/r03_d_tests.log
python3 bench.py --mode sam2_stream --steps 5 --warmup 2 > $O/r03_stream_d.json 2> $O/r03_stream_d.err
cd /tmp && export TMPDIR=/tmp
export RGA3_BENCH_TIMED_ONLY=1
rm -rf $O/prof_stream; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o stream -- python3 $R/bench.py --mode sam2_stream --steps 3 --warmup 2 > $O/prof_stream.log 2>&1
python3 $R/tools/frame_timeline.py $O/prof_stream --list --anchor "conv3x3s2_kernel<true>" > $O/r03_stream_frame_timeline_d.txt 2>&1
find $O/prof_stream -name "*kernel_trace.csv" -size +30M -delete
tail -8 $O/r03_d_tests.log; cut -c1-250 $O/r03_stream_d.json; python3 -c "
import json;d=json.loads(open('$O/r03_stream_d.json').read().strip().splitlines()[-1]);print(json.dumps(d['roofline'])[:900]);print(json.dumps(d['cpu_baseline'])[:600])"; tail -3 $O/r03_stream_d.err; grep -v "^  +" $O/r03_stream_frame_timeline_d.txt | head -24 | cut -c1-140
