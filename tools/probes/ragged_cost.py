#!/usr/bin/env python3
"""Cost of a ragged tile (64 rows, quarter-work loop of tiles 26 / 27) in units of a full 256 x 256 tile, measured where the ragged pairs ARE the critical path:
N = 131072 = 512 tile columns, K = 3584.  M = 256: 512 full tiles = 2 rounds of 256 CUs (2 t).  M = 320 on tile 27: 256 pairs of ragged tiles (round 0, every CU) + the
512 full tiles (rounds 1, 2): 2 c' t + 2 t, c' = cost of a ragged tile (incl. its epilogue and item overhead).  python3 tools/probes/ragged_cost.py [lib-suffix|-]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import torch  # noqa: E402

from rga3.hip import lib as _lib  # noqa: E402

suffix = sys.argv[1] if len(sys.argv) > 1 else "-"
if suffix != "-":
    _lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", f"librga3_hip_{suffix}.so")
from rga3.hip import ops  # noqa: E402


def t_us(M, N, K, tile, act="none"):
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N // 2 if act == "swiglu" else N, device="cuda", dtype=torch.bfloat16)
    best = 1e9
    for _ in range(4):
        for _ in range(2):
            ops.gemm(a, w, act=act, out=out, tile=tile)
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(5):
            ops.gemm(a, w, act=act, out=out, tile=tile)
        en.record()
        en.synchronize()
        best = min(best, st.elapsed_time(en) / 5 * 1e3)
    return best


for K in (3584, 1280):
    for act in ("none", "swiglu"):
        N = 131072
        t2 = t_us(256, N, K, 21, act)
        t320 = t_us(320, N, K, 27, act)
        t320p = t_us(320, N, K, 21, act)
        t = t2 / 2
        print(f"lib {suffix} K={K} {act:7s}: full tile t = {t:6.1f} us ({K // 64} K-tiles: {t / (K // 64) * 1e3:5.0f} ns per K-tile); M=320 tile 27 {t320:7.1f} us -> ragged tile = "
              f"{(t320 - t2) / 2 / t:5.3f} t ({(t320 - t2) / 2 / (K // 64) * 1e3:5.0f} ns per K-tile); padded (tile 21, 4 rounds) {t320p:7.1f} us", flush=True)
