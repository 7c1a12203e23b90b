#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 2400 python -m pytest tests/ -x -q -m gpu > $O/r04m_gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -4 $O/r04m_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r04m_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/r04m_smoke.log
timeout -k 10 900 python bench.py > $O/r04m_headline.json 2> $O/r04m_headline.err; echo "headline rc $?"
python3 - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r04m_headline.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('whole_forward_frac'), d.get('config',{}).get('variants'))
P
