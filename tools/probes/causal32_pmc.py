"""Only the causal decoder attention (attn_causal32_kernel and, with impl = 4, the general kernel) at S = 2112, 28 / 4 heads: the target of rocprofv3 --pmc passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
S, Hq, Hkv, D = 2112, 28, 4, 128
torch.manual_seed(0)
qkv = torch.randn(S, Hq + 2 * Hkv, D, device="cuda").to(torch.bfloat16)
q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:]
cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
for impl in (0, 4):
    for _ in range(6):
        ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, True, impl=impl)
torch.cuda.synchronize()
print("ok")
