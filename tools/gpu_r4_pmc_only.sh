#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pu; i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d /tmp/pu/p$i -- python3 $R/tools/pmc_pipe_util.py run > $O/r04_pmc_pipe_p$i.log 2>&1; echo "pipe pass $i rc $?"
done
python3 $R/tools/pmc_pipe_util.py sum /tmp/pu $O/r04_pmc_pipe_util.json > $O/r04_pmc_pipe_util.txt 2>&1; tail -60 $O/r04_pmc_pipe_util.txt
