#!/usr/bin/env python3
"""attn32.hip against the 16-row kernels on the model's long non-causal shapes: Hiera-L global attention (8 frames x 4096 tokens, 8 heads x 72) and the ViT's full-attention
blocks (8 segments x 1024, 16 heads x 80).  impl=1 keeps the first-generation path, impl=0 routes to attn32 when max_k is given."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import torch
from rga3.hip import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for name, nseg, L, H, D in (("hiera global x8 frames", 8, 4096, 8, 72), ("vit full x8 segments", 8, 1024, 16, 80), ("hiera global x1", 1, 4096, 8, 72)):
    qkv = (torch.randn(nseg * L, 3 * H, D, device=dev) * 0.7).to(torch.bfloat16)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    cu = (torch.arange(nseg + 1, dtype=torch.int64) * L).to(torch.int32).to(dev)
    fl = 4.0 * nseg * L * L * D * H
    a = ops.attn_varlen(q, k, v, cu, cu, L, D ** -0.5, max_k=L)
    b = ops.attn_varlen(q, k, v, cu, cu, L, D ** -0.5)          # no max_k: the former route... (self-attention derives it) -> compare against impl 2 below
    c = ops.attn_varlen(q, k, v, cu, cu, L, D ** -0.5, impl=2, max_k=L)
    err = ((a.float() - c.float()).norm() / c.float().norm()).item()
    t_new = t(lambda: ops.attn_varlen(q, k, v, cu, cu, L, D ** -0.5, max_k=L))
    t_old = t(lambda: ops.attn_varlen(q, k, v, cu, cu, L, D ** -0.5, impl=2, max_k=L))
    cu2 = cu.clone()        # distinct cu objects and no max_k: the dispatcher cannot know the key range -> the round-2 production route (8 waves x 2 x 16 rows)
    t_prod = t(lambda: ops.attn_varlen(q, k, v, cu, cu2, L, D ** -0.5))
    print(f"   round-2 production route: {t_prod:.3f} ms = {fl / t_prod / 1e9:.0f} TFLOP/s")
    print(f"{name}: attn32 {t_new:.3f} ms = {fl / t_new / 1e9:.0f} TFLOP/s   16-row {t_old:.3f} ms = {fl / t_old / 1e9:.0f} TFLOP/s   rel diff {err:.2e}", flush=True)
