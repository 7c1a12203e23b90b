#!/bin/bash
# round 5: dK/dV kernel with prefetched Q / dO tiles: parity + same-box A/B against the previous library; stage-2 MLP with 6 vs 18 fragments read ahead
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest -x -q -m gpu tests/test_kernels_gpu.py tests/test_fullsize_parity_gpu.py -k "attn or attention" > $O/r05h_attn_tests.log 2>&1; echo "attn tests rc $?"; tail -4 $O/r05h_attn_tests.log | cut -c1-600
for rep in 1 2 3; do
 for v in old -; do
  echo "lib $v:"; timeout -k 10 300 python3 tools/probes/run_with_lib.py $v tools/attn_bwd_bench.py 2>&1 | grep "S="
 done
done
for rep in 1 2; do
 for v in pre18 -; do
  echo "lib $v:"; timeout -k 10 600 python3 tools/probes/run_with_lib.py $v tools/ab_hiera_mlp.py 3 2>&1 | grep -v amdgpu | tail -1
 done
done
