#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_sam2_kernels_gpu.py -x -q -m gpu -k memattn 2>&1 | tail -15 > $O/r03_memattn_test.log
python3 -m pytest tests/test_sam2_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu -k "not mask_decoder_sam2_l" 2>&1 | tail -15 > $O/r03_sam2_tests.log
python3 tools/decoder_fullsize_grad.py 1.0 firm > $O/r03_decoder_grad_firm.log 2>&1
python3 bench.py --mode sam2_stream --steps 5 --warmup 2 > $O/r03_stream_b.json 2> $O/r03_stream_b.err
cd /tmp && export TMPDIR=/tmp
export RGA3_BENCH_TIMED_ONLY=1
rm -rf $O/prof_stream; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o stream -- python3 $R/bench.py --mode sam2_stream --steps 3 --warmup 2 > $O/prof_stream.log 2>&1
python3 $R/tools/frame_timeline.py $O/prof_stream --list --anchor "conv3x3s2_kernel<true>" > $O/r03_stream_frame_timeline_b.txt 2>&1
find $O/prof_stream -name "*kernel_trace.csv" -size +30M -delete
tail -8 $O/r03_memattn_test.log; tail -8 $O/r03_sam2_tests.log; head -30 $O/r03_decoder_grad_firm.log; tail -8 $O/r03_decoder_grad_firm.log; cut -c1-330 $O/r03_stream_b.json; tail -3 $O/r03_stream_b.err; grep -v "^  +" $O/r03_stream_frame_timeline_b.txt | head -30 | cut -c1-140
