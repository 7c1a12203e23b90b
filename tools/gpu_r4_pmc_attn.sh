#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa; i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pa/p$i -- python3 $R/tools/probes/causal32_pmc.py > $O/r04_pmc_attn_p$i.log 2>&1
  echo "pass $i rc $?"
done
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pa/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        key = 'causal32' if 'causal32' in n else ('general' if 'attn_fwd_kernel' in n else None)
        if key: acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, cs in acc.items():
    print('==', key)
    for k, v in sorted(cs.items()):
        print(f"  {k:34s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
P
