#!/bin/bash
# round 3: row-chain kernels of the memory-attention layer -- parity, then the stream bench with and without them (interleaved)
cd /root/repo
timeout 900 python -m pytest tests/test_sam2_kernels_gpu.py -x -q -k "memory_layer or memattn" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_sam2_gpu.py tests/test_fullsize_parity_gpu.py -x -q -k "memory or stream or session or propagat or video" 2>&1 | tail -3
for i in 1 2; do
for f in "" "--no-rowchain"; do
timeout 600 python bench.py --mode sam2_stream $f --steps 5 --warmup 2 2>&1 | F="$f" python3 -c "
import sys, json, os
ok = False
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('STREAM [%s]' % os.environ['F'], d['value'], d['ms_per_step']); ok = True
    last = l
if not ok: print('FAILED', last)"
done
done
