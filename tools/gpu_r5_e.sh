#!/bin/bash
# round 5, fifth GPU call: same-box A/B of the library before / after the three-phase 192-row loop (forward + training step, interleaved), and the fused-MLP ablation
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python3 tools/probes/hm_ablate.py > $O/r05e_hm_ablate.log 2>&1; echo "ablate rc $?"; grep -v amdgpu $O/r05e_hm_ablate.log | tail -5
for MODE in forward train_full; do
for rep in 1 2; do
 for v in old -; do
  timeout -k 10 600 python3 tools/probes/bench_with_lib.py $v --mode $MODE --steps 20 --warmup 5 --no-cpu-baseline > $O/r05e_ab_${MODE}_${v}_$rep.json 2> $O/r05e_ab_${MODE}_${v}_$rep.err
  python3 - $O/r05e_ab_${MODE}_${v}_$rep.json $v $MODE <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(sys.argv[3], sys.argv[2], d['value'], d['ms_per_step'], r.get('whole_forward_frac'), r.get('frac'), r.get('gemm_ms_per_step'))
except Exception as e: print('parse', e)
P
 done
done
done
