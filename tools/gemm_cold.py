"""Warm vs cold operands: the same GEMM timed on ONE weight matrix (stays in the 256-MiB Infinity Cache between calls) and rotating over
enough distinct weight / activation buffers that every call streams them from HBM -- what the model sees in situ.
  python tools/gemm_cold.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

SHAPES = [(8192, 3840, 1280, (21, 4, 31)), (8192, 1280, 1280, (4, 21)), (8192, 6912, 1280, (22, 21, 32)), (8192, 1280, 3456, (4, 22, 32)),
          (2112, 4608, 3584, (4, 21, 31, 32)), (2112, 3584, 3584, (3, 11, 22, 31, 32)), (2112, 37888, 3584, (22, 31, 32)), (2112, 3584, 18944, (22, 32)),
          (2112, 152064, 3584, (21, 31, 32))]
res = []
for (M, N, K, tiles) in SHAPES:
    nbuf = max(2, min(int(1.2e9 / (N * K * 2)) + 1, 24))
    ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    nact = max(2, int(0.6e9 / (M * K * 2)) + 1)
    acts = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(nact)]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for t in tiles:
        line = {"shape": [M, N, K], "tile": t}
        for mode in ("warm", "cold_w", "cold_aw"):
            iters = 4 * nbuf
            def run(i):
                w = ws[0] if mode == "warm" else ws[i % nbuf]
                a = acts[i % nact] if mode == "cold_aw" else acts[0]
                ops.gemm(a, w, out=out, tile=t)
            for i in range(nbuf):
                run(i)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for i in range(iters):
                run(i)
            en.record(); en.synchronize()
            us = st.elapsed_time(en) / iters * 1e3
            line[mode + "_us"] = round(us, 1)
            line[mode + "_tf"] = round(2.0 * M * N * K / us / 1e6, 1)
        print(json.dumps(line), flush=True)
        res.append(line)
    del ws, acts
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_cold.json"), "w"), indent=1)
