#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
python3 bench.py --mode forward --steps 20 --warmup 5 --no-cpu-baseline > $O/r05cc_forward.json 2> $O/r05cc_forward.err; python3 -c "
import json; d=json.loads(open('$O/r05cc_forward.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['whole_forward_frac'], r.get('board'), r.get('traffic'))"
python3 bench.py > $O/r05cc_headline.json 2> $O/r05cc_headline.err; python3 -c "
import json; d=json.loads(open('$O/r05cc_headline.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['whole_forward_frac'], r.get('board'), r.get('traffic'))"; tail -2 $O/r05cc_headline.err
