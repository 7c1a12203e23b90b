#!/bin/bash
# K = 576 family: time per tiling + PMC passes (L2 hit rate, fetch / write bytes, MFMA busy, VALU share)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-a}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/tools/probes/k576_probe.py time $O/r04_k576_time_$TAG.json > $O/r04_k576_time_$TAG.log 2>&1; echo "time rc $?"
if [ "$2" != "nopmc" ]; then
rm -rf /tmp/pk; i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pk/p$i -- python3 $R/tools/probes/k576_probe.py pmc > $O/r04_k576_pmc_p$i.log 2>&1
  echo "pmc pass $i rc $?"
done
python3 $R/tools/probes/k576_probe.py sum /tmp/pk $O/r04_k576_pmc_$TAG.json > $O/r04_k576_pmc_$TAG.txt 2>&1
fi
tail -40 $O/r04_k576_time_$TAG.log
