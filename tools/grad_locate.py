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
    def cmp(name, got, got_grad, ref, ref_grad):
        print(f"{name:10s} value rel {rl(got.detach(), ref.detach()):.4f}   grad rel {rl(got_grad, ref_grad):.4f}   |grad| {float(ref_grad.float().norm()):.3e}", flush=True)

    h = ri["upscaled"].shape[2] // 4
    cmp("high", d["high"], d["high"].grad, ri["high"].reshape(-1, 1, 1024, 1024)[sl], ri["high"].grad.reshape(-1, 1, 1024, 1024)[sl])
    cmp("masks", d["masks"], d["masks"].grad, ri["masks"][sl], ri["masks"].grad[sl])
    cmp("hyper", d["hyper"], d["hyper"].grad, ri["hyper"][sl], ri["hyper"].grad[sl])
    tok2map = lambda t: t.float().view(T, 4 * h, 4 * h, -1).permute(0, 3, 1, 2)     # product [T*16hw, C/8] token-major -> [T, C/8, 4h, 4w]
    cmp("up", tok2map(d["up"].detach()), tok2map(d["up"].grad), ri["upscaled"][sl], ri["upscaled"].grad[sl])
    cmp("hs", d["hs"], d["hs"].grad, ri["hs"][sl], ri["hs"].grad[sl])
    s2t = lambda t: t.float().view(T, -1, t.shape[-1])
    cmp("src", s2t(d["src"].detach()), s2t(d["src"].grad), ri["src"][sl].flatten(2).transpose(1, 2), ri["src"].grad[sl].flatten(2).transpose(1, 2))
    cmp("tokens", s2t(d["tokens"].detach()), s2t(d["tokens"].grad), ri["tokens"][sl], ri["tokens"].grad[sl])
    for k in ("loss", "mask_bce_loss", "mask_dice_loss"):
        print(k, float(out[k]), float(ro[k]))


if __name__ == "__main__":
    main()
