#!/usr/bin/env python3
"""coexec_probe for the memory-attention kernels: victims memlayer_rows (LayerNorm -> 256 x 768 product -> RoPE; out-proj + residual -> LayerNorm -> q product) and
memattn_cross (partials) at the SAM2-L frame shapes, each repeated on stream A beside an aggressor looping on stream B; every output compared bit for bit with the solo
result.  python3 tools/probes/coexec_probe2.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda"
g = torch.Generator().manual_seed(1)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
nq, nk, D = 4096, 28736, 256
x, lnw, lnb = rn(nq, D), rn(D, sc=0.1) + 1, rn(D, sc=0.1)
wqkv, bqkv = rn(3 * D, D, sc=0.06), rn(3 * D, sc=0.1)
wq, bq, wo, bo = rn(D, D, sc=0.06), rn(D, sc=0.1), rn(D, D, sc=0.06), rn(D, sc=0.1)
cos = torch.rand(nq, 128, generator=g).to(dev).float().contiguous()
sin = torch.rand(nq, 128, generator=g).to(dev).float().contiguous()
o_att = rn(nq, D)
mq, mk, mm = rn(nq, D), rn(nk, D), rn(nk, 64)
wov, bov = rn(D, 64, sc=0.1), rn(D, sc=0.1)
big_a, big_w = rn(8192, 1024), rn(4096, 1024, sc=0.03)
q3 = rn(nq, 3, D)
cu = torch.tensor([0, nq], dtype=torch.int32, device=dev)


def v_rows1():
    return ops.memlayer_rows(x, (lnw, lnb), 1e-5, w2=wqkv, b2=bqkv, rope=(cos, sin), rope_cols=2 * D)[2]


def v_rows2():
    r = ops.memlayer_rows(x, (lnw, lnb), 1e-5, a=o_att, w1=wo, b1=bo, w2=wq, b2=bq, rope=(cos, sin), rope_cols=D)
    return torch.cat([r[0], r[2]], 1)


def v_cross():
    po, pml, ns = ops.memattn_cross(mq, mk, mm, D ** -0.5, partials=True)
    return torch.cat([po.clone(), pml.clone()])


def v_rows3():
    parts = ops.memattn_cross(mq, mk, mm, D ** -0.5, partials=True)
    r = ops.memlayer_rows(x, (lnw, lnb), 1e-5, partials=parts, w1=wov, b1=bov, want_t=True)
    return torch.cat([r[0], r[1]], 1)


def v_selfattn():
    return ops.attn_varlen(q3[:, 0:1], q3[:, 1:2], q3[:, 2:3], cu, cu, nq, D ** -0.5)


VICT = {"memlayer_rows (LN -> qkv -> rope)": v_rows1, "memlayer_rows (out-proj + res -> LN -> q -> rope)": v_rows2, "memattn_cross partials": v_cross,
        "memattn_cross + memlayer_rows (merge -> Wo Wv + res -> LN)": v_rows3, "attn_varlen 4096 x 4096, one 256-d head": v_selfattn}
AGG = dict(VICT)
AGG["gemm tile 20 (8192 x 4096 x 1024)"] = lambda: ops.gemm(big_a, big_w, tile=20)
AGG["gemm tile 13 (8192 x 4096 x 1024)"] = lambda: ops.gemm(big_a, big_w, tile=13)
AGG["gemm tile 5 (8192 x 4096 x 1024)"] = lambda: ops.gemm(big_a, big_w, tile=5)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for vn, vf in VICT.items():
    solo = vf()
    torch.cuda.synchronize()
    again = vf()
    torch.cuda.synchronize()
    print(f"{vn}: solo twice {'equal' if torch.equal(solo, again) else 'DIFFER'}", flush=True)
    for an, af in AGG.items():
        outs = []
        with torch.cuda.stream(sb):
            for _ in range(iters // 3 + 8):
                af()
        with torch.cuda.stream(sa):      # (victims that use the stream-keyed scratch run on their own stream: the aggressor has its own)
            for _ in range(iters):
                outs.append(vf())
        torch.cuda.synchronize()
        bad = [o for o in outs if not torch.equal(o, solo)]
        if bad:
            cnt = [int((o != solo).sum()) for o in bad]
            print(f"   beside {an}: {len(bad)} / {iters} outputs differ (elements: min {min(cnt)}, max {max(cnt)}; max abs {max(float((o.float() - solo.float()).abs().max()) for o in bad):.3e})", flush=True)
    print(f"{vn}: done", flush=True)
