"""A/B of the stream-K slab hand-off (library built with make AB=1): RGA3_SK_PLAIN=1 = plain stores + release fence (round 3), 0 = write-through stores."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
dev = "cuda"
def rn(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
for (M, N, K, act, res, t) in [(2112, 3584, 18944, "none", True, 22), (2112, 3584, 18944, "none", True, 32), (2112, 3584, 3584, "none", True, 22), (2112, 37888, 3584, "swiglu", False, 22),
                               (8192, 6912, 1280, "swiglu", False, 22), (8192, 1280, 3456, "none", True, 22), (2112, 152064, 3584, "none", False, 22)]:
    nw = max(2, min(int(0.7e9 / (N * K * 2)) + 1, 32))
    ws = [rn(N, K, sc=0.03) for _ in range(nw)]
    a = rn(M, K); r = rn(M, N) if res else None
    ts = {"0": [], "1": []}
    for rnd in range(6):
        for mode in ("1", "0"):
            os.environ["RGA3_SK_PLAIN"] = mode
            for i in range(nw): ops.gemm(a, ws[i % nw], residual=r, act=act, tile=t)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for i in range(2 * nw): ops.gemm(a, ws[i % nw], residual=r, act=act, tile=t)
            en.record(); en.synchronize()
            ts[mode].append(st.elapsed_time(en) / (2 * nw) * 1e3)
    print(M, N, K, act, "tile", t, "plain+fence %.1f us   write-through %.1f us" % (statistics.median(ts["1"]), statistics.median(ts["0"])), "timeouts", ops.gemm_stream_k_timeouts(), flush=True)
