"""Fixed per-tile cost of a GEMM tiling: time(K) = a + b*K at fixed M, N.  python tools/gemm_kscan.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
M, N = 8192, 4096   # 512 tiles of 256x256 = exactly two rounds on 256 CUs
for tile in (20, 21):
    for act, bias in (("none", False),):
        pts = []
        for K in (256, 512, 1024, 2048, 4096):
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
            b = torch.randn(N, device="cuda").to(torch.bfloat16) if bias else None
            c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            for _ in range(5):
                ops.gemm(a, w, bias=b, act=act, out=c, tile=tile)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(20):
                ops.gemm(a, w, bias=b, act=act, out=c, tile=tile)
            en.record(); en.synchronize()
            pts.append((K, st.elapsed_time(en) / 20 * 1e3))
        (k0, t0), (k1, t1) = pts[1], pts[-1]
        slope = (t1 - t0) / (k1 - k0)
        print(f"tile {tile} act={act} bias={bias}: " + " ".join(f"K={k}:{t:.1f}us" for k, t in pts) + f" | per-64-K-tile {slope*64/2:.3f} us/round, fixed per round {(t0 - slope*k0)/2:.2f} us")
