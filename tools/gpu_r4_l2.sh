#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl; i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d /tmp/pl/p$i -- python3 $R/tools/pmc_pipe_util.py run > $O/r04_l2_p$i.log 2>&1; echo "pass $i rc $?"
done
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pl/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if 'rga3::' not in n: continue
        acc[n[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, cs in acc.items():
    m = {k: sum(v)/len(v) for k, v in cs.items()}
    line = f"{key:72s}"
    if 'TCC_HIT_sum' in m: line += f" L2 hit {m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum']):.3f} req {m['TCC_HIT_sum']+m['TCC_MISS_sum']:.3e}"
    if 'FETCH_SIZE' in m: line += f" fetch(x2) {m['FETCH_SIZE']*2048/1e6:.0f} MB"
    if 'GRBM_GUI_ACTIVE' in m: line += f" cyc {m['GRBM_GUI_ACTIVE']/8:.0f} mfma {m['SQ_VALU_MFMA_BUSY_CYCLES']/(m['GRBM_GUI_ACTIVE']/8*1024):.3f} valu_busy {m['SQ_ACTIVE_INST_VALU']*4/(m['GRBM_GUI_ACTIVE']/8*1024):.3f} valu_insts {m['SQ_INSTS_VALU']:.3e}"
    print(line)
P
