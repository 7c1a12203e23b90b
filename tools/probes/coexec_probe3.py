#!/usr/bin/env python3
"""coexec_probe2's failing victim -- memlayer_rows (LayerNorm -> 256 x 768 product -> RoPE) -- beside more aggressors, to tell WHAT about the GEMM tiles 5 / 13 disturbs it:
LDS-DMA users that can / cannot share a CU with its 70 KiB of LDS, LDS users without LDS-DMA, kernels without LDS.  python3 tools/probes/coexec_probe3.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = "cuda"
g = torch.Generator().manual_seed(1)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
nq, D = 4096, 256
x, lnw, lnb = rn(nq, D), rn(D, sc=0.1) + 1, rn(D, sc=0.1)
wqkv, bqkv = rn(3 * D, D, sc=0.06), rn(3 * D, sc=0.1)
wq, bq = rn(D, D, sc=0.06), rn(D, sc=0.1)
cos = torch.rand(nq, 128, generator=g).to(dev).float().contiguous()
sin = torch.rand(nq, 128, generator=g).to(dev).float().contiguous()
big_a, big_w = rn(8192, 1024), rn(4096, 1024, sc=0.03)
sm_a, sm_w = rn(8192, 64), rn(4096, 64, sc=0.1)
res = rn(8192, 4096)
qkvh = rn(32768, 24, 72)
cuw = torch.arange(0, 32768 + 1, 256, dtype=torch.int32, device=dev)
xm = rn(65536, 144)
from rga3.hip.ops import fold_layernorm
wf2, colc2, bf2 = fold_layernorm(rn(576, 144, sc=0.05), rn(576), rn(144) + 1, rn(144, sc=0.1))
w2m, b2m = rn(144, 576, sc=0.05), rn(144)
S, Hq, Hkv = 2112, 28, 4
qkvc = rn(S, Hq + 2 * Hkv, 128)
cuc = torch.tensor([0, S], dtype=torch.int32, device=dev)
ew_a, ew_b = rn(8192, 4096), rn(8192, 4096)

VICT = {
    "memlayer_rows LN -> qkv (768) -> rope": lambda: ops.memlayer_rows(x, (lnw, lnb), 1e-5, w2=wqkv, b2=bqkv, rope=(cos, sin), rope_cols=2 * D)[2],
    "memlayer_rows LN -> q (256) -> rope": lambda: ops.memlayer_rows(x, (lnw, lnb), 1e-5, w2=wq, b2=bq, rope=(cos, sin), rope_cols=D)[2],
    "memlayer_rows LN only": lambda: ops.memlayer_rows(x, (lnw, lnb), 1e-5, want_t=True)[1],
    "memlayer_rows LN -> q (256), no rope": lambda: ops.memlayer_rows(x, (lnw, lnb), 1e-5, w2=wq, b2=bq)[2],
    "memlayer_rows LN -> q (256), no rope, no bias": lambda: ops.memlayer_rows(x, (lnw, lnb), 1e-5, w2=wq)[2],
    "memlayer_rows LN -> q (256) -> rope, t written": lambda: torch.cat(ops.memlayer_rows(x, (lnw, lnb), 1e-5, w2=wq, b2=bq, rope=(cos, sin), rope_cols=D, want_t=True)[1:], 1),
}
if len(sys.argv) > 2:
    VICT = {k: v for k, v in VICT.items() if sys.argv[2] in k}
AGG0 = AGG = {
    "gemm tile 5, K = 1024": lambda: ops.gemm(big_a, big_w, tile=5),
    "gemm tile 5, K = 64 (one K-tile)": lambda: ops.gemm(sm_a, sm_w, tile=5),
    "gemm tile 5 + residual": lambda: ops.gemm(big_a, big_w, residual=res, tile=5),
    "gemm tile 13, K = 1024": lambda: ops.gemm(big_a, big_w, tile=13),
    "gemm tile 12, K = 1024": lambda: ops.gemm(big_a, big_w, tile=12),
    "gemm tile 3, K = 1024": lambda: ops.gemm(big_a, big_w, tile=3),
    "gemm tile 7 (128 x 192, three stages: 120 KiB), K = 1024": lambda: ops.gemm(big_a, big_w, tile=7),
    "gemm tile 8 (128 x 128, three stages: 96 KiB), K = 1024": lambda: ops.gemm(big_a, big_w, tile=8),
    "attn_win 256-token windows (ds_write staging, 2 x 36 KiB)": lambda: ops.attn_varlen(qkvh[:, :8], qkvh[:, 8:16], qkvh[:, 16:], cuw, cuw, 256, 72 ** -0.5, max_k=256),
    "attn_causal32 (LDS-DMA, 128 KiB)": lambda: ops.attn_varlen(qkvc[:, :Hq], qkvc[:, Hq:Hq + Hkv], qkvc[:, Hq + Hkv:], cuc, cuc, S, 128 ** -0.5, causal=True),
    "hiera_mlp144 (ds_write staging)": lambda: ops.hiera_mlp(xm, wf2, colc2, bf2, w2m, b2m, 1e-6),
    "elementwise add (no LDS)": lambda: ops.add(ew_a, ew_b),
}
if len(sys.argv) > 2:
    AGG = {k: v for k, v in AGG0.items() if any(t in k for t in ("tile 5, K = 1024", "tile 7", "tile 13", "tile 12", "attn_win"))}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for vn, vf in VICT.items():
    solo = vf()
    torch.cuda.synchronize()
    for an, af in AGG.items():
        outs = []
        with torch.cuda.stream(sb):
            for _ in range(iters // 3 + 8):
                af()
        with torch.cuda.stream(sa):
            for _ in range(iters):
                outs.append(vf())
        torch.cuda.synchronize()
        bad = [o for o in outs if not torch.equal(o, solo)]
        msg = "ok"
        if bad:
            cnt = [int((o != solo).sum()) for o in bad]
            msg = f"{len(bad)} / {iters} outputs differ (elements: min {min(cnt)}, max {max(cnt)})"
        print(f"{vn} | beside {an}: {msg}", flush=True)
