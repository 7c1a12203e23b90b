"""stream-K shapes of the forward (tile 22 / 32 / 21), cold weights."""
import os, sys, json, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
dev = "cuda"
def rn(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
for (M, N, K, act, res, tiles) in [(2112, 3584, 18944, "none", True, (22, 32, 21)), (2112, 3584, 3584, "none", True, (22, 3, 12)), (2112, 37888, 3584, "swiglu", False, (22, 21, 32)),
                                   (8192, 1280, 3456, "none", True, (22, 31, 32)), (8192, 6912, 1280, "swiglu", False, (22, 21)), (2112, 4608, 3584, "none", False, (22, 32, 31, 4))]:
    nw = max(2, min(int(0.7e9 / (N * K * 2)) + 1, 32))
    ws = [rn(N, K, sc=0.03) for _ in range(nw)]
    a = rn(M, K); r = rn(M, N) if res else None
    ref = ops.gemm(a, ws[0], residual=r, act=act, tile=10).float()
    out = {}
    for t in tiles:
        o = ops.gemm(a, ws[0], residual=r, act=act, tile=t).float()
        err = float((o - ref).norm() / ref.norm())
        ts = []
        for rnd in range(5):
            for i in range(nw): ops.gemm(a, ws[i % nw], residual=r, act=act, tile=t)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for i in range(2 * nw): ops.gemm(a, ws[i % nw], residual=r, act=act, tile=t)
            en.record(); en.synchronize()
            ts.append(st.elapsed_time(en) / (2 * nw) * 1e3)
        out[t] = (round(statistics.median(ts), 1), f"{err:.1e}")
    print(M, N, K, act, out, "timeouts", ops.gemm_stream_k_timeouts(), flush=True)
