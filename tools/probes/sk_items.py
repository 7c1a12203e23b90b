#!/usr/bin/env python3
"""Where a persistent-GEMM item spends its time (AB build: hipcc ... -DRGA3_AB -c csrc/gemm_bf16.hip linked with the product objects into rga3-release_amd/librga3_hip_ab.so; RGA3_SK_DBG=1): wave 0 of every workgroup stamps s_memtime at
0 item start | 1 first K-tile landed + barrier | 2 main loop done | 3 next item set up + its prologue issued | 4 epilogue done | 5 item end."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
os.environ["RGA3_SK_DBG"] = "1"
from rga3.hip import lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", "librga3_hip_ab.so")
from rga3.hip import ops
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
rn = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(bf)
cases = [("LLM gate-up swiglu", 2112, 37888, 3584, "swiglu", False), ("LLM gate-up none", 2112, 37888, 3584, "none", False), ("Hiera fc2", 65536, 576, 2304, "none", True),
         ("ViT gate-up swiglu", 8192, 6912, 1280, "swiglu", False), ("K576 plain 2304", 65536, 2304, 576, "none", False), ("K576 gelu 2304", 65536, 2304, 576, "gelu", False)]
ws = ops.gemm_workspace(torch.device(dev))
for name, M, N, K, act, res in cases:
    x, w = rn(M, K), rn(N, K, scale=0.02)
    r = rn(M, N) if res else None
    for _ in range(2):
        ops.gemm(x, w, act=act, residual=r, tile=21)
    torch.cuda.synchronize()
    ws[4096:4096 + 256 * 64 * 8].zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(x, w, act=act, residual=r, tile=21); e1.record(); torch.cuda.synchronize()
    st = ws[4096:4096 + 256 * 64 * 8].view(torch.int64).view(256, 8, 8).cpu().double()
    ok = st[:, :, 5] > 0
    nit = ok.sum(1).double()
    d = {}
    d["wait first tiles"] = (st[:, :, 1] - st[:, :, 0])[ok]
    d["main loop"] = (st[:, :, 2] - st[:, :, 1])[ok]
    d["next setup + prologue issue"] = (st[:, :, 3] - st[:, :, 2])[ok]
    d["epilogue"] = (st[:, :, 4] - st[:, :, 3])[ok]
    d["end barrier"] = (st[:, :, 5] - st[:, :, 4])[ok]
    nxt = (st[:, 1:, 0] - st[:, :-1, 5])[ok[:, 1:]]
    tot = (st[:, :, 5] - st[:, :, 0])[ok]
    nk = K // 64
    print(f"{name}: {M}x{N}x{K} {e0.elapsed_time(e1)*1e3:.1f} us, items/WG {float(nit.mean()):.2f}, s_memtime ticks per item {float(tot.mean()):.0f} (100 MHz ticks => us x100? see ratio)")
    for k, v in d.items():
        print(f"    {k:30s} mean {float(v.mean()):9.1f}  ({100*float(v.mean())/float(tot.mean()):5.1f} %)")
    print(f"    between items {float(nxt.mean()) if nxt.numel() else 0:9.1f};  main loop per K-tile {float(d['main loop'].mean())/nk:7.2f} ticks; kernel span {float((st[:, :, 5].max() - st[:, :, 0][st[:, :, 0] > 0].min())):.0f} ticks")
