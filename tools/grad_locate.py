#!/usr/bin/env python3
"""Where does the mask-path gradient error enter?  Compares the GRADIENTS OF INTERMEDIATE ACTIVATIONS of the mask head (mask logits, upscaled embedding,
hyper-network outputs, transformer outputs, input tokens) between the HIP path and the fp32 oracle on the tiny joint fixture.  Diagnostic only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import unigr as U  # noqa: E402
from tests.qwen_tiny import oracle_cfg  # noqa: E402
from tests.unigr_tiny import CASES, SEG, gold, make_batch, params, sam_cfg  # noqa: E402
from tools.grad_bisect import build, rl, to_dev  # noqa: E402

G = gold()


def main():
    import rga3.model.sam2 as PS2
    P, PS = params(G)
    for k in P:
        if ("text_hidden_fcs" in k) or k in ("lm_head.weight", "model.embed_tokens.weight"):
            P[k].requires_grad_(True)
    for k in PS:
        if k.startswith("sam_mask_decoder."):
            PS[k].requires_grad_(True)
    b = make_batch(CASES["10"], seed=3)      # one [SEG] sample (the product runs SAM2 per sample; the oracle batches all frames)
    ri = {}
    ro = U.model_forward(P, PS, oracle_cfg(), sam_cfg(), b, (1.0, 0.5, 2.0), SEG, internals=ri)
    ro["loss"].backward()
    m = build()
    PS2._DEBUG = {}
    out = m(**to_dev(b), inference=False)
    out["loss"].backward()
    d = PS2._DEBUG
    T = 2
    Bh = ri["masks"].shape[0]
    print("oracle frames", Bh, "product masks", tuple(d["masks"].shape))
    sl = slice(0, T)     # frames of sample 0 (the only one with [SEG])
    def cmp(name, got, ref):
        print(f"{name:10s} value rel {rl(got.detach(), ref.detach()):.4f}   grad rel {rl(got.grad, ref.grad):.4f}   |grad| {float(ref.grad.float().norm()):.3e}")
    h = ri["upscaled"].shape[2] // 4
    cmp("masks", d["masks"], ri["masks"][sl])
    cmp("hyper", d["hyper"], ri["hyper"][sl])
    up_ref = ri["upscaled"]
    got_up = d["up"]
    class V:   # product [T*16hw, C/8] token-major -> [T, C/8, 4h, 4w]
        pass
    g = V(); g.grad = got_up.grad.float().view(T, 4 * h, 4 * h, -1).permute(0, 3, 1, 2)
    gv = got_up.detach().float().view(T, 4 * h, 4 * h, -1).permute(0, 3, 1, 2)
    r = V(); r.grad = up_ref.grad[sl]
    print(f"{'up':10s} value rel {rl(gv, up_ref[sl].detach()):.4f}   grad rel {rl(g.grad, r.grad):.4f}   |grad| {float(r.grad.norm()):.3e}")
    cmp("hs", d["hs"], ri["hs"][sl])
    src_ref = ri["src"]       # [B, hw, C] in the oracle before the view? (two_way_transformer returns [B, hw, C])
    sg = d["src"].grad.float().view(T, -1, d["src"].shape[-1])
    sr = src_ref.grad[sl] if src_ref.grad is not None else None
    if sr is not None:
        sr = sr.reshape(T, sr.shape[1], -1) if sr.dim() == 3 else sr.flatten(2).transpose(1, 2)
        print(f"{'src':10s} grad rel {rl(sg, sr):.4f}   |grad| {float(sr.norm()):.3e}")
    tg = d["tokens"].grad.float().view(T, -1, d["tokens"].shape[-1])
    print(f"{'tokens':10s} grad rel {rl(tg, ri['tokens'].grad[sl]):.4f}   |grad| {float(ri['tokens'].grad[sl].norm()):.3e}")
    for k in ("loss", "mask_bce_loss", "mask_dice_loss"):
        print(k, float(out[k]), float(ro[k]))


if __name__ == "__main__":
    main()
