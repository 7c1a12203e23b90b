#!/usr/bin/env python3
"""Race / correctness screen for the three-phase 192-row main loop (tiles 31 / 32 of rga3_gemm_bf16): a sync-structure edit makes a NEW template
(cdna_hip_programming.md 5: "screen it for races over many runs at several sizes").  Tile 31 must be BIT-IDENTICAL to tile 21 (same per-element summation order);
tile 32 (stream-K tail) equal to rounding and identical run to run.  Shapes: the model's M = 2112 products, ragged M / N, K from one K-tile up, every epilogue kind.
python3 tools/probes/tile31_screen.py [repeats]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SHAPES = [(2112, 4608, 3584, "none", False), (2112, 3584, 3584, "none", True), (2112, 37888, 3584, "swiglu", False), (2112, 3584, 18944, "none", True),
          (2112, 18944, 3584, "none", False), (2112, 3584, 37888, "none", False), (2112, 8192, 3584, "gelu", False), (2112, 152064, 3584, "none", False),
          (192, 256, 64, "none", False), (192, 256, 128, "none", True), (192, 256, 192, "relu", False), (385, 700, 448, "none", True), (2000, 1000, 3584, "gelu", True),
          (4160, 4608, 3584, "none", False), (4160, 3584, 18944, "none", True), (577, 512, 6400, "none", False)]
bad = 0
for M, N, K, act, res in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    nout = N // 2 if act == "swiglu" else N
    r = torch.randn(M, nout, device="cuda", generator=g).to(torch.bfloat16) if res else None
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    ref = ops.gemm(a, w, bias, residual=r, act=act, tile=21)
    ref32 = None
    n31 = n32 = 0
    worst = 0.0
    for i in range(reps):
        y = ops.gemm(a, w, bias, residual=r, act=act, tile=31)
        if not torch.equal(y, ref):
            n31 += 1
            worst = max(worst, float((y.float() - ref.float()).abs().max()))
        z = ops.gemm(a, w, bias, residual=r, act=act, tile=32)
        if ref32 is None:
            ref32 = z
            e = float(((z.float() - ref.float()).norm() / ref.float().norm()))
            if e > 2e-3:
                n32 += 1
        elif not torch.equal(z, ref32):
            n32 += 1
    bad += n31 + n32
    print(f"M={M:<6d} N={N:<7d} K={K:<6d} {act:7s} res={int(res)}: tile 31 != tile 21 in {n31}/{reps} runs (max abs diff {worst:.3g}), tile 32 off / not reproducible in {n32}/{reps}", flush=True)
torch.cuda.synchronize()
print("stream-K give-ups:", ops.gemm_stream_k_timeouts(a.device))
print("SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
