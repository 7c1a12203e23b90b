"""Attention forward rates on a few shapes (TFLOP/s counts 4*Lq*Lk*D*H, halved for causal).  python tools/attn_rates.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
def run(B, L, Hq, Hkv, D, causal, impl=0):
    q = torch.randn(B * L, Hq, D, device="cuda").to(torch.bfloat16)
    k = torch.randn(B * L, Hkv, D, device="cuda").to(torch.bfloat16)
    v = torch.randn(B * L, Hkv, D, device="cuda").to(torch.bfloat16)
    cu = torch.arange(0, (B + 1) * L, L, dtype=torch.int32, device="cuda")
    o = torch.empty_like(q)
    f = lambda: ops.attn_varlen(q, k, v, cu, cu, L, D ** -0.5, causal, out=o, impl=impl)
    for _ in range(3): f()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(10): f()
    en.record(); en.synchronize()
    ms = st.elapsed_time(en) / 10
    fl = 4.0 * B * L * L * D * Hq * (0.5 if causal else 1.0)
    print(f"B{B} L{L} Hq{Hq} Hkv{Hkv} D{D} causal={causal} impl={impl}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF/s", flush=True)
for impl in (0,):
    run(16, 2048, 64, 8, 128, False, impl)
    run(16, 2048, 64, 8, 128, True, impl)
    run(1, 2112, 28, 4, 128, True, impl)
    run(1, 4160, 28, 4, 128, True, impl)
    run(16, 4096, 16, 16, 64, False, impl)
    run(8, 1024, 16, 16, 80, False, impl)
    run(128, 64, 16, 16, 80, False, impl)
