import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
for (M, N, K) in [(2112, 4608, 3584), (2112, 3584, 3584), (2112, 37888, 3584), (2112, 3584, 18944), (4160, 37888, 3584), (4160, 3584, 18944), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    qa, sa = ops.quant_fp8_rows(a)
    qw, sw = ops.quant_fp8_rows(w)
    c = ops.gemm_fp8(qa, sa, qw, sw)
    def t(fn, n=12):
        fn()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(n):
            fn()
        en.record(); en.synchronize()
        return st.elapsed_time(en) / n
    ms8 = t(lambda: ops.gemm_fp8(qa, sa, qw, sw, out=c))
    msq = t(lambda: ops.quant_fp8_rows(a))
    ms16 = min(t(lambda tl=tl: ops.gemm(a, w, out=c, tile=tl)) for tl in (20, 21, 22, 12))
    print(f"{M}x{N}x{K}: fp8 {2.0*M*N*K/ms8/1e9:.0f} TF ({ms8*1e3:.0f} us) + quant A {msq*1e3:.1f} us | bf16 best {2.0*M*N*K/ms16/1e9:.0f} TF ({ms16*1e3:.0f} us) | speedup incl. quant {ms16/(ms8+msq):.2f}x")
