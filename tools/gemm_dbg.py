import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 21
for (M, N, K) in [(256, 256, 256), (256, 256, 64), (512, 512, 1024)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    ref = ops.gemm(a, w, tile=10).float()
    out = ops.gemm(a, w, tile=tile).float()
    bad = (out - ref).abs() > 1e-3 * ref.abs().max()
    print(M, N, K, "bad frac", bad.float().mean().item())
    rows = bad.any(1).nonzero().flatten().tolist()
    cols = bad.any(0).nonzero().flatten().tolist()
    print(" bad rows", rows[:40], len(rows))
    print(" bad cols", cols[:40], len(cols))
    if bad.any():
        i, j = bad.nonzero()[0].tolist()
        print(" first bad", i, j, out[i, j].item(), ref[i, j].item())
        # is the wrong value equal to some other ref entry?
        m = (ref == out[i, j]).nonzero()
        print(" value found at", m[:4].tolist())
