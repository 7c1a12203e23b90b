#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bn
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bn -o bn -- python3 $R/tools/blas_kernel_names.py > $O/r04_blas_names.log 2>&1
f=$(find /tmp/bn -name '*kernel_stats.csv' | head -1)
python3 - $f <<'P' > $O/r04_blas_kernel_names.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'Cijk' in r['Name'] or 'gemm' in r['Name'].lower():
        print(r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us', r['Name'])
P
cat $O/r04_blas_kernel_names.txt | cut -c1-400
