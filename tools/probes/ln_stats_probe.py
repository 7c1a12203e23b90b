#!/usr/bin/env python3
"""GB/s of rga3_layernorm_stats at the Hiera-L shapes (8 frames): stage 3 (32768 x 576), stage 4 (8192 x 1152), stage 2 (131072 x 288), stage 1 (524288 x 144)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
for rows, dim in ((32768, 576), (65536, 576), (8192, 1152), (131072, 288), (524288, 144)):
    xs = [torch.randn(rows, dim, device="cuda").to(torch.bfloat16) for _ in range(4)]
    for x in xs: ops.layernorm_stats(x, 1e-6)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(40): ops.layernorm_stats(xs[i % 4], 1e-6)
    en.record(); en.synchronize()
    us = st.elapsed_time(en) / 40 * 1e3
    print(f"rows {rows:7d} dim {dim:5d}: {us:6.1f} us  {rows * dim * 2 / us / 1e3:7.1f} GB/s", flush=True)
