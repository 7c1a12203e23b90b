#!/usr/bin/env python3
"""Row-chain kernels of the memory-attention layer (csrc/memlayer.hip) timed beside the launches they replace, at the stream's shapes (4096 rows).
python3 tools/memlayer_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402


def t_us(f, n=20, reps=10):
    """device time per call: n calls captured in one hipGraph (no host launch cost between them), replayed reps times"""
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                f()
        g.replay()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(reps):
            g.replay()
        en.record()
        en.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    return st.elapsed_time(en) / (n * reps) * 1e3


def main():
    dev, M = "cuda", 4096
    r = lambda *s, sc=1.0: (torch.randn(*s) * sc).to(torch.bfloat16).to(dev)
    x, a, a2 = r(M, 256), r(M, 256), r(M, 768)
    gam, bet = r(256), r(256, sc=0.1)
    ang = torch.rand(M, 128) * 6.28
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    cos2, sin2 = cos.repeat(1, 2).contiguous(), sin.repeat(1, 2).contiguous()
    wqkv, bqkv = r(768, 256, sc=0.06), r(768, sc=0.1)
    wo, bo, wq, bq = r(256, 256, sc=0.06), r(256, sc=0.1), r(256, 256, sc=0.06), r(256, sc=0.1)
    wov, bov = r(256, 64, sc=0.1), r(256, sc=0.1)
    qq, kk, mm = r(M, 256), r(28736, 256), r(28736, 64)

    def un1():
        y = ops.gemm(ops.layernorm(x, gam, bet, 1e-5), wqkv, bqkv)
        ops.rope_axial_(y[:, :512], cos2, sin2, M)

    def un2():
        xx = ops.gemm(a, wo, bo, residual=x)
        q = ops.gemm(ops.layernorm(xx, gam, bet, 1e-5), wq, bq)
        ops.rope_axial_(q, cos, sin, M)

    pm = ops.memattn_cross(qq, kk, mm, 256 ** -0.5)
    parts = ops.memattn_cross(qq, kk, mm, 256 ** -0.5, partials=True)

    def un3():
        xx = ops.gemm(pm, wov, bov, residual=x)
        ops.layernorm(xx, gam, bet, 1e-5)

    print(f"floors: layernorm {t_us(lambda: ops.layernorm(x, gam, bet, 1e-5)):6.1f}  gemm 256->768 {t_us(lambda: ops.gemm(x, wqkv, bqkv)):6.1f}  gemm 256->256 {t_us(lambda: ops.gemm(x, wq, bq)):6.1f}  "
          f"gemm 64->256 + res {t_us(lambda: ops.gemm(pm, wov, bov, residual=x)):6.1f}  rope 512 {t_us(lambda: ops.rope_axial_(a2[:, :512], cos2, sin2, M)):6.1f}  rope 256 {t_us(lambda: ops.rope_axial_(a, cos, sin, M)):6.1f} us")
    print(f"norm only (row-chain kernel): {t_us(lambda: ops.memlayer_rows(x, (gam, bet), 1e-5, want_t=True)):6.1f} us;  norm -> q (256) -> rope: {t_us(lambda: ops.memlayer_rows(x, (gam, bet), 1e-5, w2=wq, b2=bq, rope=(cos, sin), rope_cols=256)):6.1f} us"
          f";  out + res -> norm: {t_us(lambda: ops.memlayer_rows(x, (gam, bet), 1e-5, a=a, w1=wo, b1=bo, want_t=True)):6.1f} us")
    print(f"norm -> qkv -> rope         : fused {t_us(lambda: ops.memlayer_rows(x, (gam, bet), 1e-5, w2=wqkv, b2=bqkv, rope=(cos, sin), rope_cols=512)):7.1f} us   separate {t_us(un1):7.1f} us")
    print(f"out + res -> norm -> q, rope: fused {t_us(lambda: ops.memlayer_rows(x, (gam, bet), 1e-5, a=a, w1=wo, b1=bo, w2=wq, b2=bq, rope=(cos, sin), rope_cols=256)):7.1f} us   separate {t_us(un2):7.1f} us")
    print(f"merge -> WoWv + res -> norm : fused {t_us(lambda: ops.memlayer_rows(x, (gam, bet), 1e-5, partials=parts, w1=wov, b1=bov, want_t=True)):7.1f} us   separate (without the merge) {t_us(un3):7.1f} us")
    for (m_, n_, k_) in ((9, 256, 256), (9, 2048, 256), (9, 256, 2048)):
        ta, tw, tb = r(m_, k_), r(n_, k_, sc=0.05), r(n_)
        print(f"token rows {m_} x {n_} x {k_}: tile 41 {t_us(lambda: ops.gemm(ta, tw, tb, tile=41)):6.1f} us   128 x 128 tiles {t_us(lambda: ops.gemm(ta, tw, tb, tile=12)):6.1f} us")
    mw = [(r(256, 256, sc=0.06), r(256, sc=0.1), r(256, 256, sc=0.06), r(256, sc=0.1), r(32, 256, sc=0.06), r(32, sc=0.1)) for _ in range(6)]
    toks = r(1, 9, 256)
    print(f"six 3-layer MLPs on token rows (mlp3_rows): {t_us(lambda: ops.mlp3_rows([(toks[0, i], 9 * 256, mw[i], False) for i in range(6)], 1)):7.1f} us")
    dx, dw_, db = r(4096, 256), r(256, 1, 7, 7, sc=0.1), r(256, sc=0.1)
    print(f"dwconv 7x7, 64 x 64 x 256   : {t_us(lambda: ops.dwconv7x7(dx, dw_, db, 1, 64, 64)):7.1f} us")
    print(f"cross attention             : partials only {t_us(lambda: ops.memattn_cross(qq, kk, mm, 256 ** -0.5, partials=True)):7.1f} us   with merge {t_us(lambda: ops.memattn_cross(qq, kk, mm, 256 ** -0.5)):7.1f} us")


if __name__ == "__main__":
    main()
