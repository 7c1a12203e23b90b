#!/bin/bash
# round 5: launch point of the SAM2-encoder prefetch inside the backward: sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_unigr_gpu.py -k prefetch > $O/r05l_unigr.log 2>&1; echo "unigr rc $?"; tail -3 $O/r05l_unigr.log | cut -c1-600
for L in 0 4 7 10 14; do
timeout -k 10 900 python3 bench.py --mode train_full --steps 20 --warmup 5 --no-cpu-baseline --sam-prefetch-layer $L > $O/r05l_train_$L.json 2> $O/r05l_train_$L.err
python3 - $O/r05l_train_$L.json $L <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('layer', sys.argv[2], d['value'], d['ms_per_step'], {k: v for k, v in d['config']['variants'].items() if 'prefetch' in k})
except Exception as e: print('parse', e)
P
done
