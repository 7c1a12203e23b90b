"""Cost of the RMSNorm-fold epilogues: each producer / consumer product of the forward with and without rms_out / rms_in (cold weights, interleaved)."""
import os, sys, json, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
dev = "cuda"
def rn(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
CASES = [("o_proj   producer", 2112, 3584, 3584, "none", True, "out"), ("down     producer", 2112, 3584, 18944, "none", True, "out"),
         ("qkv      consumer", 2112, 4608, 3584, "none", False, "in"), ("gate-up  consumer", 2112, 37888, 3584, "swiglu", False, "in"),
         ("vit proj producer", 8192, 1280, 1280, "none", True, "out"), ("vit fc2  producer", 8192, 1280, 3456, "none", True, "out"),
         ("vit qkv  consumer", 8192, 3840, 1280, "none", False, "in"), ("vit g-up consumer", 8192, 6912, 1280, "swiglu", False, "in")]
for name, M, N, K, act, res, kind in CASES:
    nw = max(2, min(int(0.7e9 / (N * K * 2)) + 1, 32))
    ws = [rn(N, K, sc=0.03) for _ in range(nw)]
    a = rn(M, K); r = rn(M, N) if res else None
    bias = rn(N) if not res else None
    sums = torch.zeros(M, dtype=torch.int64, device=dev)
    sums_in = (torch.rand(M, device=dev) * K * 2 ** 20).to(torch.int64)
    ops.gemm(a, ws[0], bias, residual=r, act=act)      # tune
    def run(i, on):
        kw = {}
        if on and kind == "out": kw["rms_out"] = sums
        if on and kind == "in": kw["rms_in"] = (sums_in, K, 1e-6)
        ops.gemm(a, ws[i % nw], bias, residual=r, act=act, **kw)
    t = {True: [], False: []}
    for rnd in range(6):
        for on in (False, True):
            for i in range(nw): run(i, on)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            n = 2 * nw
            for i in range(n): run(i, on)
            en.record(); en.synchronize()
            t[on].append(st.elapsed_time(en) / n * 1e3)
    print(f"{name}  {M}x{N}x{K}: plain {statistics.median(t[False]):7.1f} us   rms {statistics.median(t[True]):7.1f} us   (+{statistics.median(t[True]) - statistics.median(t[False]):.1f})", flush=True)
