#!/bin/bash
# Round-6 GPU jobs, ONE script: tools/gpu_r6.sh <section> [<section> ...]   (run through gpurun; artefacts under gpurun_out/, copied to profiles/r06_* by hand)
#   newtests   the tests added this round (fast)
#   encprof    SAM2-L image encoder alone, 16 frames: wall time + rocprofv3 kernel stats
#   suite      the driver's two commands (pytest -m gpu, smoke)
#   headline   python bench.py --gpus 1   (driver's command)
#   forward    python bench.py --mode forward
#   stream     configs[3], 1 and 4 objects
#   fp8        configs[4]
#   fp8prof    rocprofv3 kernel statistics of one configs[4] optimizer step
#   timeline   rocprofv3 kernel trace of one training step -> step timeline
#   probe:<file.py>[:arg,arg,...]   python3 tools/probes/<file.py> arg arg ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
for s in "$@"; do
  echo "=== section $s"
  case $s in
    newtests)
      timeout -k 10 1500 python3 -m pytest -x -q -m gpu tests/test_sam2_gpu.py tests/test_unigr_gpu.py -k "concurrent or pending or two_objects or prefetch" > $O/r06_newtests.log 2>&1; echo "rc $?"; tail -5 $O/r06_newtests.log | cut -c1-400
      timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_fullsize_parity_gpu.py -k "concurrent_slot" >> $O/r06_newtests.log 2>&1; echo "rc $?"; tail -5 $O/r06_newtests.log | cut -c1-400 ;;
    encprof)
      NF=16 timeout -k 10 600 python3 tools/sam2_encoder_probe.py 5 > $O/r06_encoder_probe.log 2>&1; grep "ms per" $O/r06_encoder_probe.log
      NF=16 RGA3_TUNE_SAVE=/tmp/enc_tuner.json timeout -k 10 600 python3 tools/sam2_encoder_probe.py 1 > /dev/null 2>&1    # the tilings first: the profiled run then holds no trial launches
      rm -rf $O/encprof; NF=16 RGA3_TUNE_LOAD=/tmp/enc_tuner.json timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encprof -o enc -- python3 tools/sam2_encoder_probe.py 8 > $O/r06_encoder_prof.log 2>&1
      f=$(find $O/encprof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_encoder_kernel_stats.csv && head -40 $O/r06_encoder_kernel_stats.csv | cut -c1-200
      rm -rf $O/encprof ;;
    lnsum)
      timeout -k 10 1200 python3 -m pytest -x -q -m gpu tests/test_kernels_gpu.py -k "layernorm" > $O/r06_lnsum_tests.log 2>&1; echo "rc $?"; tail -5 $O/r06_lnsum_tests.log | cut -c1-400
      timeout -k 10 1200 python3 -m pytest -x -q -m gpu tests/test_sam2_gpu.py tests/test_fullsize_parity_gpu.py -k "image_encoder or hiera" >> $O/r06_lnsum_tests.log 2>&1; echo "rc $?"; tail -5 $O/r06_lnsum_tests.log | cut -c1-400
      NF=16 AB=_LN_SUMS timeout -k 10 600 python3 tools/sam2_encoder_probe.py 5 > $O/r06_encoder_ab_lnsums.log 2>&1; grep "ms per\|A/B" $O/r06_encoder_ab_lnsums.log ;;
    attnwin)
      timeout -k 10 1200 python3 -m pytest -x -q -m gpu tests/test_kernels_gpu.py tests/test_sam2_kernels_gpu.py -k "attn or attention" > $O/r06_attn_tests.log 2>&1; echo "rc $?"; tail -3 $O/r06_attn_tests.log | cut -c1-400
      timeout -k 10 600 python3 tools/hiera_attn_probe.py > $O/r06_hiera_attn_probe.log 2>&1; cat $O/r06_hiera_attn_probe.log | cut -c1-200 ;;
    train)
      timeout -k 10 2400 python3 -m pytest -x -q -m gpu tests/test_train_gpu.py tests/test_kernels_gpu.py -k "dropout or training or attn_window or lora" > $O/r06_train_tests.log 2>&1; echo "rc $?"; tail -4 $O/r06_train_tests.log | cut -c1-400 ;;
    suite)
      timeout -k 10 3000 python3 -m pytest tests/ -x -q -m gpu > $O/r06_gpu_tests.log 2>&1; echo "suite rc $?"; tail -3 $O/r06_gpu_tests.log | cut -c1-300
      timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/r06_smoke.log ;;
    headline)
      python3 bench.py --gpus 1 > $O/r06_bench_headline.json 2> $O/r06_bench_headline.err; tail -c 400 $O/r06_bench_headline.json ;;
    forward)
      python3 bench.py --mode forward > $O/r06_bench_forward.json 2> $O/r06_bench_forward.err; tail -c 400 $O/r06_bench_forward.json ;;
    stream)
      python3 bench.py --mode sam2_stream --steps 5 --warmup 2 > $O/r06_bench_sam2_stream.json 2> $O/r06_bench_sam2_stream.err; tail -c 300 $O/r06_bench_sam2_stream.json
      python3 bench.py --mode sam2_stream --objects 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_sam2_stream_4obj.json 2> $O/r06_bench_sam2_stream_4obj.err; tail -c 300 $O/r06_bench_sam2_stream_4obj.json ;;
    fp8)
      python3 bench.py --mode lora_fp8 --steps 3 --warmup 1 > $O/r06_bench_lora_fp8.json 2> $O/r06_bench_lora_fp8.err; tail -c 300 $O/r06_bench_lora_fp8.json ;;
    frameline)
      rm -rf /tmp/pfs; (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pfs -o st -- python3 $R/bench.py --mode sam2_stream --steps 2 --warmup 1 --no-cpu-baseline --no-board > $O/r06_prof_stream.log 2>&1)
      python3 tools/frame_timeline.py /tmp/pfs --list --anchor "${ANCHOR:-sam_select_kernel}" --group-ms 0.3 --frame ${FRAME:-50} > $O/r06_stream_frame_timeline.txt 2>&1; head -150 $O/r06_stream_frame_timeline.txt | cut -c1-150
      cut -d, -f8- $(find /tmp/pfs -name '*kernel_trace.csv' | head -1) | grep -o 'rga3::[a-z_0-9]*' | sort | uniq -c | sort -rn | head -50
      rm -rf /tmp/pfs ;;
    atencensus)
      RGA3_BENCH_ATEN_CENSUS=$O/r06_train_step_aten_census.txt python3 bench.py --mode train_full --steps 3 --warmup 3 --no-cpu-baseline --no-board > /dev/null 2> $O/r06_atencensus.err; cat $O/r06_train_step_aten_census.txt | cut -c1-230 ;;
    fp8prof)
      rm -rf /tmp/p8; (cd /tmp && timeout -k 10 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -o f8 -- python3 $R/bench.py --mode lora_fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-board > $O/r06_prof_lora_fp8.log 2>&1)
      f=$(find /tmp/p8 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_bench_lora_fp8_kernel_stats.csv && head -45 $O/r06_bench_lora_fp8_kernel_stats.csv | cut -c1-220
      rm -rf /tmp/p8 ;;
    timeline)
      rm -rf /tmp/pt; timeout -k 10 1200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o tr -- python3 bench.py --mode train_full --steps 4 --warmup 3 --no-cpu-baseline --no-board > $O/r06_timeline_run.log 2>&1
      python3 tools/step_timeline.py /tmp/pt --bin-ms 5 --from-ms 0 --to-ms 1000 --exclude gemm_ > $O/r06_train_step_timeline.txt 2>&1; head -70 $O/r06_train_step_timeline.txt | cut -c1-160
      rm -rf /tmp/pt ;;
    probe:*)
      IFS=: read -r _ file pargs <<< "$s"
      timeout -k 10 2400 python3 tools/probes/$file ${pargs//,/ } > $O/r06_${file%.py}.log 2>&1; echo "rc $?"; tail -40 $O/r06_${file%.py}.log | cut -c1-300 ;;
    *) echo "unknown section $s" ;;
  esac
done
