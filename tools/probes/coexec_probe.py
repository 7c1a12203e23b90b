#!/usr/bin/env python3
"""Does a GEMM tiling give the same bits when another kernel runs beside it?  A victim product (fixed operands, one forced tiling) is launched repeatedly on stream A
while an aggressor loops on stream B; every victim output is compared bit for bit with the solo result.  python3 tools/probes/coexec_probe.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = "cuda"
g = torch.Generator().manual_seed(1)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
SHAPES = [(4096, 256, 256, True), (4096, 768, 256, False), (28736, 1024, 64, False), (4096, 256, 64, True), (4096, 2048, 256, False), (16384, 64, 256, False)]
mq, mk, mm = rn(4096, 256), rn(28736, 256), rn(28736, 64)
big_a, big_w = rn(8192, 1024), rn(4096, 1024, sc=0.03)
AGG = {
    "memattn_cross": lambda: ops.memattn_cross(mq, mk, mm, 256 ** -0.5, partials=True),
    "gemm tile 20 (8192 x 4096 x 1024)": lambda: ops.gemm(big_a, big_w, tile=20),
    "gemm tile 13 (8192 x 4096 x 1024)": lambda: ops.gemm(big_a, big_w, tile=13),
    "gemm tile 5 (8192 x 4096 x 1024)": lambda: ops.gemm(big_a, big_w, tile=5),
}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for (M, N, K, res) in SHAPES:
    a, w, b = rn(M, K), rn(N, K, sc=K ** -0.5), rn(N, sc=0.1)
    r = rn(M, N) if res else None
    for tile in (5, 13, 12, 3, 20):
        if tile == 5 and N % 8:
            continue
        solo = ops.gemm(a, w, b, residual=r, tile=tile)
        torch.cuda.synchronize()
        for name, agg in AGG.items():
            bad = 0
            outs = []
            with torch.cuda.stream(sb):
                for _ in range(iters // 4 + 8):
                    agg()
            with torch.cuda.stream(sa):
                for _ in range(iters):
                    outs.append(ops.gemm(a, w, b, residual=r, tile=tile))
            torch.cuda.synchronize()
            bad = sum(0 if torch.equal(o, solo) else 1 for o in outs)
            if bad:
                worst = max(float((o.float() - solo.float()).abs().max()) for o in outs)
                print(f"({M}, {N}, {K}, res={res}) tile {tile:2d} beside {name}: {bad} / {iters} outputs differ from the solo result, max abs {worst:.3e}", flush=True)
        print(f"({M}, {N}, {K}, res={res}) tile {tile:2d}: done", flush=True)
