cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf /tmp/pq1 /tmp/pq2 /tmp/pq3
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/pq1 -- python3 $R/tools/probes/swiglu_quant_probe.py > $O/r06_quant_pmc_p1.log 2>&1; echo "p1 rc $?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pq2 -- python3 $R/tools/probes/swiglu_quant_probe.py > $O/r06_quant_pmc_p2.log 2>&1; echo "p2 rc $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pq3 -- python3 $R/tools/probes/swiglu_quant_probe.py > $O/r06_quant_pmc_p3.log 2>&1; echo "p3 rc $?"
mkdir -p /tmp/pqa; cp -r /tmp/pq1 /tmp/pq2 /tmp/pq3 /tmp/pqa/ 2>/dev/null
python3 $R/tools/probes/pmc_sum_by_kernel.py /tmp/pqa quant > $O/r06_quant_pmc.txt 2>&1; cat $O/r06_quant_pmc.txt | cut -c1-260
