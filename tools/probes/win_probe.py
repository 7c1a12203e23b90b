#!/usr/bin/env python3
"""Window attention (attn_win_kernel): Hiera stage-3 windows (256 tokens, 8 x 72), stage-2 (4 x 72), 64-token windows and the ViT's 64-token windows (16 x 80, RoPE on
load): time; DBG_LIB=old for the previous build."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import lib as _lib
if os.environ.get("DBG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", "librga3_hip_%s.so" % os.environ["DBG_LIB"])
from rga3.hip import ops
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
def timeit(fn, n=6, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for name, T, win, H, D in [("Hiera s3 windows (16 frames)", 65536, 256, 8, 72), ("Hiera s2 windows", 262144, 256, 4, 72), ("Hiera s3, 64-token windows", 65536, 64, 8, 72),
                           ("ViT windows", 8192, 64, 16, 80), ("D = 64 windows", 32768, 256, 8, 64)]:
    qkv = (torch.randn(T, 3 * H, D, device=dev)).to(bf)
    cu = torch.arange(0, T + 1, win, dtype=torch.int32, device=dev)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out = ops.attn_varlen(q, k, v, cu, cu, win, D ** -0.5, causal=False)
    sh = lambda x: x[:2 * win].reshape(2, win, H, D).transpose(1, 2).float()
    ref = torch.nn.functional.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(2 * win, H, D)
    err = float((out[:2 * win].float() - ref).abs().max())
    us = timeit(lambda: ops.attn_varlen(q, k, v, cu, cu, win, D ** -0.5, causal=False))
    fl = 4.0 * T * win * D * H
    line = f"{name:30s} T={T} win={win} H={H} D={D}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  max err {err:.4f}"
    if D == 80:
        pos = torch.arange(T, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, D // 2, dtype=torch.float32) / (D // 2)))[None, :]
        cos, sin = torch.cat([pos.cos(), pos.cos()], 1).contiguous().to(dev), torch.cat([pos.sin(), pos.sin()], 1).contiguous().to(dev)
        us2 = timeit(lambda: ops.attn_varlen_rope(q, k, v, cu, cu, win, D ** -0.5, cos, sin, causal=False, rope_k=True))
        line += f"   with RoPE on load {us2:8.1f} us"
    print(line, flush=True)
