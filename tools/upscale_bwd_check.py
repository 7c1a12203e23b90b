#!/usr/bin/env python3
"""Stage-by-stage gradient check of the mask decoder's upscaling path (ConvT 2x2 -> +feat_s1 -> LayerNorm2d -> GELU -> ConvT 2x2 -> +feat_s0 -> GELU)
on the HIP autograd nodes against an fp32 torch chain on the same bf16 inputs.  Diagnostic (tools/grad_locate.py showed d(src) 34 % off while d(up) is 2 %)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from rga3.hip import autograd as AG  # noqa: E402

dev = torch.device("cuda:0")


def rl(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def run(Fn, h, C, gscale, seed, common_mode):
    g = torch.Generator().manual_seed(seed)
    R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    src = R(Fn * h * h, C).to(torch.bfloat16)
    w1, b1 = R(C, C // 4, 2, 2, sc=C ** -0.5).to(torch.bfloat16), R(C // 4, sc=0.02).to(torch.bfloat16)
    lw, lb = (1 + 0.1 * R(C // 4)).to(torch.bfloat16), R(C // 4, sc=0.02).to(torch.bfloat16)
    w2, b2 = R(C // 4, C // 8, 2, 2, sc=(C // 4) ** -0.5).to(torch.bfloat16), R(C // 8, sc=0.02).to(torch.bfloat16)
    s1 = R(Fn * 4 * h * h, C // 4).to(torch.bfloat16)
    s0 = R(Fn * 16 * h * h, C // 8).to(torch.bfloat16)
    dup = R(Fn * 16 * h * h, C // 8, sc=gscale)
    if common_mode:   # a gradient with a large component common to all channels (what LayerNorm backward projects out)
        dup = dup + common_mode * gscale * R(Fn * 16 * h * h, 1)
    dup = dup.to(torch.bfloat16)

    # ---- product
    P = {k: v.to(dev).requires_grad_(True) for k, v in dict(src=src, w1=w1, b1=b1, lw=lw, lb=lb, w2=w2, b2=b2).items()}
    as_lin = lambda w: w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]).contiguous()
    st = {}
    g1 = AG.linear(P["src"], as_lin(P["w1"])); st["g1"] = g1
    u1 = AG.PixelShuffleFn.apply(g1, P["b1"], s1.to(dev), Fn, h, h); st["u1"] = u1
    n1 = AG.LayerNormFn.apply(u1.contiguous(), P["lw"], P["lb"], 1e-6); st["n1"] = n1
    a1 = AG.GeluFn.apply(n1); st["a1"] = a1
    g2 = AG.linear(a1, as_lin(P["w2"])); st["g2"] = g2
    u2 = AG.PixelShuffleFn.apply(g2, P["b2"], s0.to(dev), Fn, 2 * h, 2 * h); st["u2"] = u2
    up = AG.GeluFn.apply(u2)
    for t in st.values():
        t.retain_grad()
    up.backward(dup.to(dev))

    # ---- fp32 torch reference (token-major <-> NCHW by views)
    Rf = {k: v.float().requires_grad_(True) for k, v in dict(src=src, w1=w1, b1=b1, lw=lw, lb=lb, w2=w2, b2=b2).items()}
    t2m = lambda t, H: t.view(Fn, H, H, -1).permute(0, 3, 1, 2)
    m2t = lambda m: m.permute(0, 2, 3, 1).reshape(-1, m.shape[1])
    rs = {}
    x = t2m(Rf["src"], h)
    c1 = F.conv_transpose2d(x, Rf["w1"], Rf["b1"], stride=2) + t2m(s1.float(), 2 * h); rs["u1"] = c1
    mu = c1.mean(1, keepdim=True); var = (c1 - mu).pow(2).mean(1, keepdim=True)
    n = (c1 - mu) / torch.sqrt(var + 1e-6) * Rf["lw"][:, None, None] + Rf["lb"][:, None, None]; rs["n1"] = n
    a = F.gelu(n); rs["a1"] = a
    c2 = F.conv_transpose2d(a, Rf["w2"], Rf["b2"], stride=2) + t2m(s0.float(), 4 * h); rs["u2"] = c2
    upr = F.gelu(c2)
    for t in rs.values():
        t.retain_grad()
    upr.backward(t2m(dup.float(), 4 * h))

    print(f"F={Fn} h={h} C={C} gscale={gscale:g} common_mode={common_mode}: up value rel {rl(up.detach(), m2t(upr.detach())):.4f}")
    for k in ("u2", "a1", "n1", "u1"):
        print(f"   d({k})  rel {rl(st[k].grad, m2t(rs[k].grad)):.4f}   |ref| {float(rs[k].grad.norm()):.3e}")
    print(f"   d(src) rel {rl(P['src'].grad, Rf['src'].grad):.4f}   |ref| {float(Rf['src'].grad.norm()):.3e}")
    for k in ("w1", "b1", "lw", "lb", "w2", "b2"):
        print(f"   d({k})  rel {rl(P[k].grad, Rf[k].grad):.4f}")


if __name__ == "__main__":
    run(2, 16, 256, 1.0, 0, 0.0)
    run(2, 64, 256, 1e-4, 1, 0.0)
    run(2, 64, 256, 1e-4, 2, 20.0)
