"""LoRA-side products of configs[4] (S = 4160, r = 128) and configs[2] (S = 2112): time per tiling incl. the skinny split-K form (tile 14).  python3 tools/probes/lora_shapes_probe.py"""
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


for M, N, K, res in ((4160, 128, 3584, False), (2112, 128, 3584, False), (4160, 128, 512, False), (4160, 3584, 128, True), (4160, 512, 128, True), (2112, 3584, 128, True),
                     (4160, 3584, 128, False)):
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = []
    for tile in (13, 12, 3, 5, 20, 14):
        if tile == 14 and res:
            continue
        try:
            us = t_us(lambda: ops.gemm(a, w, None, residual=r, out=out, tile=tile))
            line.append(f"{tile}:{us:6.1f}us")
        except Exception as e:  # noqa: BLE001
            line.append(f"{tile}:ERR")
    print(f"M={M:<5d} N={N:<5d} K={K:<5d} res={int(res)}  " + "  ".join(line), flush=True)
