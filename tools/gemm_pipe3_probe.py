"""Three-stage single-phase tiles (6 / 7 / 8: two K-tiles of LDS-DMA in flight) against the two-stage ones (3 / 5 / 12) and the ping-pong / persistent
family on the shapes where the model runs single-phase tiles, with COLD operands (weights and activations rotated through > 256 MiB, as in situ).
  python3 tools/gemm_pipe3_probe.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

SHAPES = [  # M, N, K, residual, act, tiles
    (2112, 3584, 3584, True, "none", (3, 6, 12, 8, 22, 31, 32)),
    (2112, 4608, 3584, False, "none", (4, 3, 6, 31, 21)),
    (8192, 1280, 1280, True, "none", (4, 31, 3, 6, 12, 8)),
    (8192, 3840, 1280, False, "none", (5, 7, 3, 6, 21, 4)),
    (8192, 1280, 3456, True, "none", (4, 3, 6, 22, 31)),
    (32768, 1728, 576, False, "none", (5, 7, 3, 6, 20)),
    (32768, 576, 576, True, "none", (5, 7, 12, 8)),
    (32768, 2304, 576, False, "gelu", (5, 7, 3, 6, 20)),
    (32768, 576, 2304, True, "none", (5, 7, 12, 8)),
    (131072, 1152, 288, False, "gelu", (5, 7, 3, 6, 12, 8)),
    (131072, 288, 1152, True, "none", (5, 7, 12, 8)),
]
res = []
for (M, N, K, has_res, act, tiles) in SHAPES:
    nw = max(2, min(int(0.6e9 / (N * K * 2)) + 1, 64))
    na = max(2, min(int(0.6e9 / (M * K * 2)) + 1, 16))
    ws = [(torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(nw)]
    acts = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(na)]
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if has_res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = ops.gemm(acts[0], ws[0], residual=r, act=act, tile=12).clone()
    line = {"shape": [M, N, K], "res": has_res, "act": act, "us": {}, "tf": {}, "equal_to_tile12": {}}
    for t in tiles:
        o = ops.gemm(acts[0], ws[0], residual=r, act=act, tile=t)
        line["equal_to_tile12"][t] = bool(torch.equal(o, ref)) if t in (3, 4, 5, 6, 7, 8, 12) else float(((o.float() - ref.float()).norm() / ref.float().norm()).item())
        iters = max(16, 2 * nw)
        for i in range(4):
            ops.gemm(acts[i % na], ws[i % nw], residual=r, act=act, out=out, tile=t)
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for i in range(iters):
            ops.gemm(acts[i % na], ws[i % nw], residual=r, act=act, out=out, tile=t)
        en.record(); en.synchronize()
        us = st.elapsed_time(en) / iters * 1e3
        line["us"][t] = round(us, 1)
        line["tf"][t] = round(2.0 * M * N * K / us / 1e6, 1)
    print(json.dumps(line), flush=True)
    res.append(line)
    del ws, acts
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
