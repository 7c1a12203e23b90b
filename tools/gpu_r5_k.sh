#!/bin/bash
# round 5: SAM2 encoder of the next sample prefetched beside the optimizer step: parity test + headline with variants
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_unigr_gpu.py > $O/r05k_unigr.log 2>&1; echo "unigr rc $?"; tail -5 $O/r05k_unigr.log | cut -c1-600
for rep in 1 2; do
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05k_headline_$rep.json 2> $O/r05k_headline_$rep.err; echo "headline rc $?"
python3 - $O/r05k_headline_$rep.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('HEAD', d['value'], d['ms_per_step'], d['config']['variants'], d['verify'].get('loss_first_last'))
except Exception as e: print('parse', e)
P
done
