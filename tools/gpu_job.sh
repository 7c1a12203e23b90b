#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "" "--refine" "" "--refine"; do T0=$(date +%s); python3 $R/bench.py --mode train_full --steps 10 --warmup 3 --no-cpu-baseline $v > $O/r4l.json 2> $O/r4l.err; python3 -c "
import json;d=json.loads(open('$O/r4l.json').read().strip().splitlines()[-1]);print('TRAIN [$v]',d['value'],d['ms_per_step'])"; echo "  wall $(( $(date +%s) - T0 )) s"; grep -c "tuner.refine" $O/r4l.err; done
