"""Run one GEMM shape repeatedly (for rocprofv3): python tools/gemm_probe.py M N K tile iters"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

M, N, K, tile, iters = (int(x) for x in sys.argv[1:6])
a = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(iters):
    ops.gemm(a, w, out=c, tile=tile)
torch.cuda.synchronize()
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(iters):
    ops.gemm(a, w, out=c, tile=tile)
en.record()
torch.cuda.synchronize()
ms = st.elapsed_time(en) / iters
print(f"M={M} N={N} K={K} tile={tile}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s")
