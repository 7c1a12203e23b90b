"""Per-call timing of the attention launches inside one SAM2-L encoder pass (8 frames).  python tools/sam2_attn_breakdown.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.model.sam2 import SAM2
from rga3.hip import ops
torch.manual_seed(1)
m = SAM2().to(torch.bfloat16).cuda().eval()
ev = []
real = ops.attn_varlen
def timed(q, k, v, cu_q, cu_k, max_q, *a, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = real(q, k, v, cu_q, cu_k, max_q, *a, **kw); e.record()
    ev.append((s, e, (q.shape[0], q.shape[1], q.shape[2], int(max_q), cu_q.numel() - 1)))
    return r
with torch.no_grad():
    x = torch.randn(8, 3, 1024, 1024, device="cuda").to(torch.bfloat16)
    m.sam2_model.forward_image(x); m.sam2_model.forward_image(x)
    ops.attn_varlen = timed
    import rga3.model.sam2 as S
    m.sam2_model.forward_image(x)
    torch.cuda.synchronize()
agg = {}
for s, e, key in ev:
    t = s.elapsed_time(e) * 1e3
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += t
tot = 0
for key, (n, t) in agg.items():
    T, H, D, mq, nseg = key
    fl = 4.0 * T * mq * D * H
    print(f"tokens {T} heads {H} D {D} window {mq} segs {nseg}: {n} calls, {t/n:.1f} us each, {fl/(t/n)/1e6:.0f} TF/s")
    tot += t
print(f"attention total {tot/1e3:.2f} ms per 8 frames")
