#!/usr/bin/env python3
"""Per-item stamps of tile 27 (persistent 256 x 256, ragged last tile row on the quarter-work loop, natural tile order) on the LLM gate | up product: how long a ragged
item's main loop takes next to the full items of its tile column (AB build: librga3_hip_ab.so, RGA3_SK_DBG=1; s_memtime ticks are 10 ns)."""
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
ws = ops.gemm_workspace(torch.device(dev))
for name, M, N, K, act in [("LLM gate-up swiglu", 2112, 37888, 3584, "swiglu"), ("LM head", 2112, 152064, 3584, "none")]:
    x, w = rn(M, K), rn(N, K, scale=0.02)
    for tile in (27, 21):
        for _ in range(2):
            ops.gemm(x, w, act=act, tile=tile)
        torch.cuda.synchronize()
        ws[4096:4096 + 256 * 64 * 8].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(x, w, act=act, tile=tile); e1.record(); torch.cuda.synchronize()
        st = ws[4096:4096 + 256 * 64 * 8].view(torch.int64).view(256, 8, 8).cpu().double()
        ok = st[:, :, 5] > 0
        t0 = st[:, :, 0][ok].min()
        ntm = (M + 255) // 256
        wg = torch.arange(256)[:, None].expand(256, 8)
        it = torch.arange(8)[None, :].expand(256, 8)
        tile_id = wg + 256 * it
        rag = (tile_id % ntm) == ntm - 1
        main = st[:, :, 2] - st[:, :, 1]
        whole = st[:, :, 5] - st[:, :, 0]
        epi = st[:, :, 4] - st[:, :, 3]
        print(f"{name} tile {tile}: {e0.elapsed_time(e1) * 1e3:.1f} us; kernel span {(st[:, :, 5].max() - t0) / 100:.1f} us")
        for lbl, m in (("full", ok & ~rag), ("ragged", ok & rag)):
            if m.any():
                print(f"    {lbl:6s} items {int(m.sum()):4d}: main loop {float(main[m].mean()) / 100:6.1f} us (min {float(main[m].min()) / 100:6.1f}, max {float(main[m].max()) / 100:6.1f}), epilogue {float(epi[m].mean()) / 100:5.1f} us, whole item {float(whole[m].mean()) / 100:6.1f} us")
        for r in range(6):
            m = ok[:, r]
            if m.any():
                s0 = (st[:, r, 0][m] - t0) / 100
                e5 = (st[:, r, 5][m] - t0) / 100
                print(f"    round {r}: {int(m.sum()):3d} items, start {float(s0.min()):6.1f} .. {float(s0.max()):6.1f} us, end {float(e5.min()):6.1f} .. {float(e5.max()):6.1f} us")
