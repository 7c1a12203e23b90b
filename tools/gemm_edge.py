import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
for (M, N, K, act) in [(2112, 37888, 3584, "swiglu"), (2112, 3584, 18944, "none"), (2112, 152064, 3584, "none"), (8192, 8192, 8192, "none"), (8192, 3840, 1280, "none")]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    line = f"{M}x{N}x{K} {act}:"
    for rnd in range(2):
        for tile in (20, 21, 22):
            c = ops.gemm(a, w, act=act, tile=tile)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(12):
                ops.gemm(a, w, act=act, out=c, tile=tile)
            en.record(); en.synchronize()
            ms = st.elapsed_time(en) / 12
            line += f" t{tile}={2.0*M*N*K/ms/1e9:.0f}"
        line += " |"
    print(line)
