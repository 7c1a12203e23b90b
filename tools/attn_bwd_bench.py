import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
for S in (2112, 4160):
    Hq, Hk, D = 28, 4, 128
    q = torch.randn(S, Hq, D, device="cuda").to(torch.bfloat16)
    k = torch.randn(S, Hk, D, device="cuda").to(torch.bfloat16)
    v = torch.randn(S, Hk, D, device="cuda").to(torch.bfloat16)
    do = torch.randn(S, Hq, D, device="cuda").to(torch.bfloat16)
    cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
    o, lse = ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, True, return_lse=True)
    def t(fn, n=10):
        fn()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(n):
            fn()
        en.record(); en.synchronize()
        return st.elapsed_time(en) / n
    mf = t(lambda: ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, True, return_lse=True))
    mb = t(lambda: ops.attn_varlen_bwd(q, k, v, o, do, lse, cu, cu, S, S, D ** -0.5, True))
    fl = 4.0 * S * S * D * Hq / 2
    print(f"S={S}: fwd {mf*1e3:.0f} us ({fl/mf/1e9:.0f} TF)  bwd {mb*1e3:.0f} us ({2.5*fl/mb/1e9:.0f} TF)")
