#!/usr/bin/env python3
"""Every GEMM tiling against an fp32 reference on the small-K / small-N shapes of the mask path (forward and backward products).  Diagnostic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import torch
from rga3.hip import ops
dev = torch.device("cuda:0")
def rel(a, b): return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()
g = torch.Generator().manual_seed(0)
shapes = [(16384, 256, 128), (65536, 128, 64), (65536, 64, 128), (16384, 256, 256), (16384, 128, 256), (36, 256, 256), (36, 2048, 256), (36, 256, 2048), (4, 256, 256), (36, 128, 256), (36, 256, 128),
          (16384, 64, 256), (262144, 32, 256), (65536, 256, 64)]
tiles = (-1, 20, 21, 22, 31, 32, 12, 13, 3, 4, 5, 14, 25)
for (M, N, K) in shapes:
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, generator=g)).to(torch.bfloat16).to(dev)
    r = (torch.randn(M, N, generator=g)).to(torch.bfloat16).to(dev)
    ref = a.float() @ w.float().t()
    out = []
    for t in tiles:
        try:
            e0 = rel(ops.gemm(a, w, tile=t), ref)
            e1 = rel(ops.gemm(a, w, b, residual=r, tile=t), ref + b.float() + r.float())
            e2 = rel(ops.gemm(a, w, out_dtype=torch.float32, tile=t), ref)
            out.append(f"{t}:{max(e0, e1):.1e}/{e2:.0e}")
        except Exception as ex:
            out.append(f"{t}:ERR")
    print((M, N, K), " ".join(out), flush=True)
