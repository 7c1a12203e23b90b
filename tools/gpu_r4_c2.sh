#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for i in 1 2; do
python3 bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r04c_fold_on_$i.json 2> $O/r04c_fold_on_$i.err; tail -3 $O/r04c_fold_on_$i.err; tail -c 600 $O/r04c_fold_on_$i.json | head -c 300; echo
RGA3_RMS_FOLD=0 python3 bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r04c_fold_off_$i.json 2> $O/r04c_fold_off_$i.err; tail -c 600 $O/r04c_fold_off_$i.json | head -c 300; echo
done
python3 - <<'P'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r04c_fold_o*_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['roofline']['whole_forward_frac'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
P
