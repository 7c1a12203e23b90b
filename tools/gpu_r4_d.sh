#!/bin/bash
# fold on / off: per-kernel statistics of the forward bench (same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in on off; do
  if [ $m = off ]; then export RGA3_RMS_FOLD=0; else export RGA3_RMS_FOLD=1; fi
  rm -rf /tmp/pf_$m; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$m -o fwd -- python3 $R/bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > $O/r04d_prof_$m.log 2>&1
  cp $(find /tmp/pf_$m -name "*kernel_stats.csv" | head -1) $O/r04d_fwd_kernel_stats_$m.csv
done
python3 - <<'P'
import csv,os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/'
for m in ('on','off'):
    rows=list(csv.DictReader(open(O+f'r04d_fwd_kernel_stats_{m}.csv')))
    print('=====',m)
    for r in rows[:16]:
        print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms  calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
P
