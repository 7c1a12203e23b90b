#!/bin/bash
# same-box A/B of two library builds: forward bench, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
MODE=${1:-forward}
for rep in 1 2; do
 for v in old -; do
  timeout -k 10 600 python3 tools/probes/bench_with_lib.py $v --mode $MODE --steps 20 --warmup 5 --no-cpu-baseline > $O/ab_${v}_$rep.json 2> $O/ab_${v}_$rep.err
  python3 - $O/ab_${v}_$rep.json $v <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d['value'], d['ms_per_step'], d['roofline'].get('whole_forward_frac'), d['roofline'].get('frac'), (d['roofline'].get('variants_ms') or {}).get('as_timed'))
except Exception as e: print('parse', e)
P
 done
done
