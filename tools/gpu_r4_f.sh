#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in on off; do
  if [ $m = off ]; then export RGA3_RMS_FOLD=0; else export RGA3_RMS_FOLD=1; fi
  RGA3_TUNE_SAVE=$O/r04f_tuner_$m.json python3 $R/bench.py --mode forward --steps 10 --warmup 3 --no-cpu-baseline > $O/r04f_fwd_$m.json 2> $O/r04f_fwd_$m.err
  export RGA3_TUNE_LOAD=$O/r04f_tuner_$m.json RGA3_BENCH_TIMED_ONLY=1
  rm -rf /tmp/pf_$m; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$m -o fwd -- python3 $R/bench.py --mode forward --steps 20 --warmup 3 --no-refine --no-cpu-baseline > $O/r04f_prof_$m.log 2>&1
  cp $(find /tmp/pf_$m -name "*kernel_stats.csv" | head -1) $O/r04f_fwd_kernel_stats_$m.csv
  unset RGA3_TUNE_LOAD RGA3_BENCH_TIMED_ONLY
done
python3 - <<'P'
import csv,os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/'
for m in ('on','off'):
    rows=list(csv.DictReader(open(O+f'r04f_fwd_kernel_stats_{m}.csv')))
    print('=====',m, sum(float(r['TotalDurationNs']) for r in rows if 'rga3' in r['Name'])/1e6/23)
    for r in rows[:18]:
        if 'rga3' in r['Name']: print(f"{float(r['TotalDurationNs'])/1e6/23:9.3f} ms/fwd  calls/fwd {int(r['Calls'])/23:6.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
P
