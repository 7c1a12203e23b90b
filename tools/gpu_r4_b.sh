#!/bin/bash
# round 4, job B: three-stage single-phase GEMM tiles, the new causal attention kernel (tests + A/B), vendor kernel names
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R && timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or gemm" > $O/r04b_tests.log 2>&1; tail -5 $O/r04b_tests.log
timeout 300 python3 tools/causal32_probe.py 2>&1 | grep -v amdgpu.ids
timeout 900 python3 tools/gemm_pipe3_probe.py $O/r04b_gemm_pipe3.json 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bn; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bn -o bn -- python3 $R/tools/blas_kernel_names.py > $O/r04b_blas_names.log 2>&1
python3 - <<'P'
import csv, glob
for f in glob.glob('/tmp/bn/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith('Cijk') or 'gemm' in r['Name'].lower():
            print(r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Name'][:400])
P
