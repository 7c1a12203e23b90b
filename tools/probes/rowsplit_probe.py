#!/usr/bin/env python3
"""M = 2112 = 6 x 256 + 3 x 192: the LLM products as two launches (rows [0, 1536) on 256-row tiles, rows [1536, 2112) on 192-row tiles) against one launch."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
rn = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(bf)
M = 2112
def timeit(fn, n=6, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort(); return ts[len(ts) // 2]
cases = [("gate-up swiglu", 37888, 3584, "swiglu", False), ("gate-up none", 37888, 3584, "none", False), ("down+res", 3584, 18944, "none", True), ("dX gate-up", 3584, 37888, "none", False),
         ("dH down", 18944, 3584, "none", False), ("qkv", 4608, 3584, "none", False), ("o+res", 3584, 3584, "none", True)]
for name, N, K, act, res in cases:
    x = rn(M, K)
    ws = [rn(N, K, scale=0.02) for _ in range(3)]     # rotate weights: cold, as inside the model
    No = N // 2 if act == "swiglu" else N
    r = rn(M, No) if res else None
    out = torch.empty(M, No, dtype=bf, device=dev)
    cnt = [0]
    def one(tile):
        cnt[0] += 1
        ops.gemm(x, ws[cnt[0] % 3], act=act, residual=r, out=out, tile=tile)
    def split(s, ta, tb):
        cnt[0] += 1
        w = ws[cnt[0] % 3]
        ops.gemm(x[:s], w, act=act, residual=None if r is None else r[:s], out=out[:s], tile=ta)
        ops.gemm(x[s:], w, act=act, residual=None if r is None else r[s:], out=out[s:], tile=tb)
    res_ = {f"tile{t}": timeit(lambda t=t: one(t)) for t in (22, 32, 21)}
    res_["1536/22 + 576/32"] = timeit(lambda: split(1536, 22, 32))
    res_["2048/22 + 64/13"] = timeit(lambda: split(2048, 22, 13))
    res_["2048/22 + 64/12"] = timeit(lambda: split(2048, 22, 12))
    res_["1792/22 + 320/4"] = timeit(lambda: split(1792, 22, 4)) if act != "swiglu" else float("nan")
    # correctness of the split against the single launch
    one(22); ref = out.clone(); cnt[0] -= 1; split(1536, 22, 32); cnt[0] -= 1
    ok = torch.equal(ref[:1536], out[:1536]) and float((ref[1536:].float() - out[1536:].float()).abs().max()) < 0.1
    print(f"{name:16s} N={N:6d} K={K:6d}: " + "  ".join(f"{k} {v:7.1f}" for k, v in res_.items()) + f"  split-ok {ok}", flush=True)
