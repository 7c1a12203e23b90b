#!/usr/bin/env python3
"""Products with a residual: time per tiling, old build (DBG_LIB=old) against the current one; results compared bit for bit with tile 13 (direct residual loads there too)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import lib as _lib
if os.environ.get("DBG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", "librga3_hip_%s.so" % os.environ["DBG_LIB"])
from rga3.hip import ops
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
rn = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(bf)
def timeit(fn, n=6, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / inner * 1e3)
    ts.sort(); return ts[len(ts) // 2]
cases = [("LLM o-proj", 2112, 3584, 3584, (3, 12, 20, 31, 5)), ("LLM down", 2112, 3584, 18944, (22, 21, 20)), ("ViT proj", 8192, 1280, 1280, (4, 3, 20, 12)), ("ViT down", 8192, 1280, 3456, (4, 3, 20, 12)),
         ("Hiera fc2", 65536, 576, 2304, (20, 21, 5, 12, 23)), ("Hiera proj", 65536, 576, 576, (5, 12, 20, 23)), ("Hiera s2 fc2", 16384, 1152, 4608, (20, 3, 5, 23)), ("ragged", 1000, 1000, 320, (3, 5, 12, 13, 20, 23))]
for name, M, N, K, tiles in cases:
    x, w, b, r = rn(M, K), rn(N, K, scale=0.02), rn(N), rn(M, N)
    ref = ops.gemm(x, w, bias=b, residual=r, tile=13)
    line = f"{name:14s} {M}x{N}x{K}:"
    for t in tiles:
        out = ops.gemm(x, w, bias=b, residual=r, tile=t)
        same = torch.equal(out, ref) or float((out.float() - ref.float()).abs().max()) < 0.07     # stream-K tilings re-associate
        line += f"  t{t} {timeit(lambda: ops.gemm(x, w, bias=b, residual=r, tile=t)):7.1f}{'' if same else ' WRONG'}"
    print(line, flush=True)
