"""Reference point only (never on the product path): what the vendor library (torch.nn.functional.linear -> hipBLASLt) reaches on the
forward's GEMM shapes on this board, next to the hand-written kernels -- shows how much of the gap to peak is the shape and how much the
kernel.  python tools/blas_reference_point.py"""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
SHAPES = [(8192, 3840, 1280), (8192, 1280, 1280), (8192, 6912, 1280), (8192, 1280, 3456), (2112, 4608, 3584), (2112, 3584, 3584),
          (2112, 37888, 3584), (2112, 3584, 18944), (2112, 152064, 3584), (8192, 8192, 8192)]
def t(fn, n):
    for _ in range(3): fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); en.synchronize()
    return st.elapsed_time(en) / n
res = []
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    nb = max(2, int(0.7e9 / (N * K * 2)) + 1)
    ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    i = [0]
    def blas():
        i[0] += 1
        torch.matmul(a, ws[i[0] % nb].t(), out=out)
    def mine():
        i[0] += 1
        ops.gemm(a, ws[i[0] % nb], out=out)
    ops.gemm(a, ws[0], out=out)   # tune
    n = 4 * nb
    tb, tm = t(blas, n), t(mine, n)
    fl = 2.0 * M * N * K
    res.append({"shape": [M, N, K], "hipblaslt_tf": round(fl / tb / 1e9, 1), "rga3_tf": round(fl / tm / 1e9, 1)})
    print(res[-1], flush=True)
    del ws
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "blas_reference_point.json"), "w"), indent=1)
