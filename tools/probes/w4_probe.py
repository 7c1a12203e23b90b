#!/usr/bin/env python3
"""Tile 28 (256 x 256 on FOUR waves: 128 x 128 wave blocks, AGPR accumulators, hand-ordered MFMA / fragment-read / LDS-DMA stream; one tile per workgroup) against the
8-wave kernels: bit-identity with tile 20 (same K order per element), reproducibility, and rates next to tiles 20 / 21 / 22 / 27.  python3 tools/probes/w4_probe.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SHAPES = [(256, 256, 64, "none", False), (256, 256, 128, "none", True), (512, 768, 192, "relu", False), (300, 700, 448, "gelu", True), (1024, 1024, 1024, "swiglu", False),
          (2112, 4608, 3584, "none", False), (2112, 37888, 3584, "swiglu", False), (2112, 3584, 18944, "none", True), (8192, 6912, 1280, "swiglu", False),
          (8192, 3840, 1280, "none", False), (8192, 1280, 3456, "none", True), (8192, 8192, 8192, "none", False), (65536, 2304, 576, "gelu", False),
          (2112, 152064, 3584, "none", False), (4096, 4096, 4096, "none", False)]


def operands(M, N, K, act, res):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    nout = N // 2 if act == "swiglu" else N
    r = torch.randn(M, nout, device="cuda", generator=g).to(torch.bfloat16) if res else None
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    return a, w, bias, r


bad = 0
for M, N, K, act, res in SHAPES:
    a, w, bias, r = operands(M, N, K, act, res)
    ref = ops.gemm(a, w, bias, residual=r, act=act, tile=20)
    nbad = 0
    info = ""
    for i in range(reps):
        z = ops.gemm(a, w, bias, residual=r, act=act, tile=28)
        if not torch.equal(z, ref):
            nbad += 1
            zf, rf = z.float(), ref.float()
            info = f" rel-L2 {float((zf - rf).norm() / rf.norm()):.2e}, {int((z != ref).sum())} elements differ, finite {bool(torch.isfinite(zf).all())}"
    bad += nbad
    nout = N // 2 if act == "swiglu" else N
    out = torch.empty(M, nout, device="cuda", dtype=torch.bfloat16)
    tiles = (20, 21, 22, 27, 28)
    best = {t: float("inf") for t in tiles}
    for rnd in range(3):
        for t in tiles:
            f = lambda: ops.gemm(a, w, bias, residual=r, act=act, out=out, tile=t)
            for _ in range(2):
                f()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(8):
                f()
            en.record()
            en.synchronize()
            best[t] = min(best[t], st.elapsed_time(en) / 8 * 1e3)
    print(f"M={M:<6d} N={N:<7d} K={K:<6d} {act:7s} res={int(res)}: tile 28 != tile 20 in {nbad}/{reps}{info};  us (TF/s)  " +
          "  ".join(f"{t}: {best[t]:7.1f} ({2.0 * M * N * K / best[t] / 1e6:5.0f})" for t in tiles), flush=True)
print("SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
