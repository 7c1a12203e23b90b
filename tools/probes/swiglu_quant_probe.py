"""configs[4]: the fused SwiGLU + e4m3 row quantisers at the decoder's shape (T = 4160, I = 18944) and the ViT's -- time per launch and bytes moved.
python3 tools/probes/swiglu_quant_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402


def t_us(f, n=30):
    for _ in range(3):
        f()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        f()
    en.record()
    en.synchronize()
    return st.elapsed_time(en) / n * 1e3


for T, I in ((4160, 18944), (16384, 3456)):
    gu = (torch.randn(T, 2 * I, device="cuda") * 1.5).to(torch.bfloat16)
    da = torch.randn(T, I, device="cuda").to(torch.bfloat16)
    us = t_us(lambda: ops.swiglu_fwd_quant(gu))
    by = T * I * (4 + 1)
    print(f"swiglu_fwd_quant T={T} I={I}: {us:7.1f} us  {by / us / 1e6:5.2f} TB/s", flush=True)
    us = t_us(lambda: ops.swiglu_bwd_quant(gu, da))
    by = T * I * (4 + 2 + 2)
    print(f"swiglu_bwd_quant T={T} I={I}: {us:7.1f} us  {by / us / 1e6:5.2f} TB/s", flush=True)
    x = torch.randn(T, 3584, device="cuda").to(torch.bfloat16)
    us = t_us(lambda: ops.quant_fp8_rows(x))
    print(f"quant_fp8_rows   T={T} K=3584: {us:7.1f} us  {T * 3584 * 3 / us / 1e6:5.2f} TB/s", flush=True)
