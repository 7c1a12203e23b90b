#!/bin/bash
# round 5: attention backward A/B: previous dK/dV kernel (oldattn), current, dQ on 4 waves x 32 rows (dqqt2)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
 for v in oldattn - dqqt2; do
  echo "lib $v:"; timeout -k 10 300 python3 tools/probes/run_with_lib.py $v tools/attn_bwd_bench.py 2>&1 | grep "S="
 done
done
