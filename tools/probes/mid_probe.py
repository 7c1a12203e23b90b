#!/usr/bin/env python3
"""Mid-size plain products of the forward on every tiling incl. 256 x 192 (tile 23): LLM qkv, ViT qkv, ViT proj, lm_head slice; cold weights (rotated)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
rn = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(bf)
def timeit(fn, n=6, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for name, M, N, K, bias in [("LLM qkv", 2112, 4608, 3584, True), ("ViT qkv", 8192, 3840, 1280, True), ("ViT fc (plain 6912)", 8192, 6912, 1280, False), ("dH-like 2112x18944x3584", 2112, 18944, 3584, False)]:
    x = rn(M, K); ws = [rn(N, K, scale=0.02) for _ in range(4)]; b = rn(N) if bias else None
    cnt = [0]
    def one(t):
        cnt[0] += 1
        ops.gemm(x, ws[cnt[0] % 4], b, tile=t)
    line = f"{name:26s}"
    for t in (23, 20, 21, 22, 31, 32, 4, 3, 12, 6):
        line += f"  t{t} {timeit(lambda: one(t)):6.1f}"
    print(line, flush=True)
