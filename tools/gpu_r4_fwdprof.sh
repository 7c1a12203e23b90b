#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python3 $R/bench.py --mode forward --steps 10 --warmup 3 --no-refine --no-cpu-baseline > $O/r04j_fwdprof.log 2>&1
f=$(find /tmp/fp -name '*kernel_stats.csv' | head -1); cp $f $O/r04j_fwd_kernel_stats.csv
python3 - $O/r04j_fwd_kernel_stats.csv <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:45]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.2f}%")
print('total ms', tot/1e6)
P
