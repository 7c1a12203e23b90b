#!/usr/bin/env python3
"""Causal attention backward at the decoder's shape (S = 2112 / 4160, 28 / 4 heads x 128): time of one rga3_attn_varlen_bwd call; DBG_LIB=<suffix> for another build."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import lib as _lib
if os.environ.get("DBG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", "librga3_hip_%s.so" % os.environ["DBG_LIB"])
from rga3.hip import ops
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
for S in (2112, 4160):
    Hq, Hkv, D = 28, 4, 128
    qkv = torch.randn(S, Hq + 2 * Hkv, D, device=dev).to(bf)
    q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:]
    cu = torch.tensor([0, S], dtype=torch.int32, device=dev)
    o, lse = ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, causal=True, return_lse=True)
    do = torch.randn(S, Hq, D, device=dev).to(bf)
    outs = None
    for _ in range(3):
        outs = ops.attn_varlen_bwd(q, k, v, o, do, lse, cu, cu, S, S, D ** -0.5, True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            outs = ops.attn_varlen_bwd(q, k, v, o, do, lse, cu, cu, S, S, D ** -0.5, True)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 4 * 1e3)
    ts.sort()
    print(f"S={S}: {ts[len(ts)//2]:.1f} us per backward call; checksums {[float(t.float().abs().sum()) for t in outs[:3]]}", flush=True)
