"""Loop one attention shape (for rocprofv3 --pmc): python tools/attn_probe.py L Hq Hkv D causal iters"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
L, Hq, Hkv, D, causal, iters = (int(x) for x in sys.argv[1:7])
q = torch.randn(L, Hq, D, device="cuda").to(torch.bfloat16)
k = torch.randn(L, Hkv, D, device="cuda").to(torch.bfloat16)
v = torch.randn(L, Hkv, D, device="cuda").to(torch.bfloat16)
cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
o = torch.empty_like(q)
for _ in range(iters):
    ops.attn_varlen(q, k, v, cu, cu, L, D ** -0.5, bool(causal), out=o)
torch.cuda.synchronize()
