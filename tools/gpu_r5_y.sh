#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for rep in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05y_plain_$rep.json 2> $O/r05y_plain_$rep.err; python3 -c "
import json; d=json.loads(open('$O/r05y_plain_$rep.json').read().strip().splitlines()[-1]); print('plain', d['ms_per_step'])"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --refine > $O/r05y_refine_$rep.json 2> $O/r05y_refine_$rep.err; python3 -c "
import json; d=json.loads(open('$O/r05y_refine_$rep.json').read().strip().splitlines()[-1]); print('refine', d['ms_per_step'])"; grep -h "tuner" $O/r05y_refine_$rep.err | cut -c1-400
done
