#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python3 -m pytest tests/test_unigr_gpu.py -x -q -m gpu -k "gradients_through_mask_path" 2>&1 | grep -v "^GRAD_ERRS\|Captured" | tail -40 | cut -c1-400 > $O/r03_e_tests.log
python3 tools/decoder_fullsize_grad.py 1.0 firm > $O/r03_decoder_grad_firm.log 2>&1
cat $O/r03_e_tests.log; head -40 $O/r03_decoder_grad_firm.log; tail -7 $O/r03_decoder_grad_firm.log
