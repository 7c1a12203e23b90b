#!/usr/bin/env python3
"""Time one GEMM shape on every tiling (HIP events, 20 launches each): python3 tools/gemm_tile_probe.py M N K [act] ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

SHAPES = [(2112, 3584, 37888, "none", False), (2112, 3584, 18944, "none", True), (2112, 3584, 4608, "none", False), (2112, 152064, 3584, "none", False),
          (65536, 576, 576, "none", True), (65536, 2304, 576, "gelu", False), (65536, 576, 2304, "none", True), (65536, 1728, 576, "none", False),
          (8192, 6912, 1280, "swiglu", False), (8192, 1280, 3456, "none", True), (8192, 3840, 1280, "none", False), (8192, 1280, 1280, "none", True),
          (262144, 1152, 288, "gelu", False), (262144, 288, 1152, "none", True), (2112, 3584, 3584, "none", True), (2112, 37888, 3584, "swiglu", False),
          (2112, 18944, 3584, "none", False), (2112, 4608, 3584, "none", False)]
for M, N, K, act, res in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    nout = N // 2 if act == "swiglu" else N
    r = torch.randn(M, nout, device="cuda").to(torch.bfloat16) if res else None
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, nout, device="cuda", dtype=torch.bfloat16)
    line = []
    for tile in (20, 21, 22, 31, 32, 12, 3, 5):
        f = lambda: ops.gemm(a, w, bias, residual=r, act=act, out=out, tile=tile)
        for _ in range(3):
            f()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(20):
            f()
        en.record()
        en.synchronize()
        us = st.elapsed_time(en) / 20 * 1e3
        line.append(f"{tile}:{2.0 * M * N * K / us / 1e6:5.0f}")
    print(f"M={M:<7d} N={N:<6d} K={K:<6d} {act:7s} res={int(res)}  TF/s by tile  " + "  ".join(line), flush=True)
