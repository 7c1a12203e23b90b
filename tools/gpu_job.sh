#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "lora" > $O/r3q_tests.log 2>&1; tail -5 $O/r3q_tests.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
