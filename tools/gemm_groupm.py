"""A/B of the tile-order group height (RGA3_GEMM_GROUPM) is per process: run once per value.  python tools/gemm_groupm.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
for (M, N, K, tile) in [(2112, 37888, 3584, 22), (2112, 3584, 18944, 22), (2112, 152064, 3584, 21), (2112, 4608, 3584, 4), (8192, 3840, 1280, 21), (8192, 6912, 1280, 22), (8192, 8192, 8192, 21)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.gemm(a, w, out=c, tile=tile)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    st.record()
    for _ in range(n):
        ops.gemm(a, w, out=c, tile=tile)
    en.record(); en.synchronize()
    ms = st.elapsed_time(en) / n
    print(f"GROUPM={os.environ.get('RGA3_GEMM_GROUPM','auto')} {M}x{N}x{K} tile {tile}: {2.0*M*N*K/ms/1e9:.0f} TF")
