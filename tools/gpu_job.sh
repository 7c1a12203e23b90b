#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --mode train_full --steps 20 --warmup 5 --no-cpu-baseline > $O/r2t_train.json 2> $O/r2t_train.err; python3 -c "
import json;d=json.loads([l for l in open('$O/r2t_train.json') if l.startswith('{')][-1]);print('TRAIN',d['value'],d['ms_per_step'],d['verify'])"
rm -rf /tmp/tt; RGA3_TUNE_LOAD=$R/profiles/r02_tuner_forward.json timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o tr -- python3 $R/bench.py --mode train_full --steps 4 --warmup 3 --no-cpu-baseline > $O/r2t_trace.log 2>&1
python3 $R/tools/trace_gaps.py /tmp/tt --last-ms 1500 > $O/r2t_gaps.txt 2>&1; head -30 $O/r2t_gaps.txt | cut -c1-180
