#!/usr/bin/env python3
"""Which op of the mask-path backward loses precision?  (median gradient error ~3x what bf16 storage alone costs.)
Runs the tiny joint model's fwd+bwd on the GPU several times, each time with ONE family of rga3.hip.autograd nodes replaced by an fp32 torch
implementation (inputs / outputs still bf16), and prints per-variant gradient errors against the fp32 oracle.  Diagnostic only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import unigr as U  # noqa: E402
from rga3.hip import autograd as AG  # noqa: E402
from tests.qwen_tiny import oracle_cfg, product_cfg_kwargs  # noqa: E402
from tests.unigr_tiny import CASES, SAM_TINY, SEG, gold, make_batch, params, sam_cfg  # noqa: E402

dev = torch.device("cuda:0")
G = gold()


def rl(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def build():
    from rga3.model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel
    cfg = UniGRConfig(train_mask_decoder=True, out_dim=256, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seg_token_idx=SEG,
                      sam_pretrained=None, sam_config=SAM_TINY, **product_cfg_kwargs())
    m = UniGRModel(cfg)
    m.initialize_sam_modules(cfg)
    P0, PS0 = params(G)
    sd = dict(P0)
    sd.update({"grounding_encoder.sam2_model." + k: v for k, v in PS0.items()})
    m.load_state_dict(sd, strict=True)
    m = m.to(torch.bfloat16).to(dev)
    for n, p in m.named_parameters():
        p.requires_grad_(("sam_mask_decoder" in n) or ("text_hidden_fcs" in n) or n in ("lm_head.weight", "model.embed_tokens.weight"))
    return m


def to_dev(b):
    out = {}
    for k, v in b.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(dev).to(torch.bfloat16) if v.is_floating_point() and k in ("pixel_values_videos", "images_sam") else v.to(dev)
        elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            out[k] = [t.to(dev) for t in v]
        else:
            out[k] = v
    return out


# ---- fp32 torch stand-ins (diagnostic)
def linear_t(a, w, bias=None, residual=None, act="none", out_f32=False):
    y = F.linear(a.float(), w.float(), None if bias is None else bias.float())
    if act == "relu":
        y = torch.relu(y.to(torch.bfloat16).float()) if not out_f32 else torch.relu(y)
    if residual is not None:
        y = y + residual.float()
    return y if out_f32 else y.to(torch.bfloat16)


class _Apply:
    def __init__(self, f):
        self.apply = f


def attn_t(q, k, v, cu_q, cu_k, max_q, max_k, scale):
    B = cu_q.numel() - 1
    H, D = q.shape[1], q.shape[2]
    qq = q.float().view(B, -1, H, D).transpose(1, 2)
    kk = k.float().view(B, -1, H, D).transpose(1, 2)
    vv = v.float().view(B, -1, H, D).transpose(1, 2)
    p = torch.softmax(qq @ kk.transpose(2, 3) * scale, -1)
    return (p @ vv).transpose(1, 2).reshape(-1, H, D).to(torch.bfloat16)


def ln_t(x, w, b, eps):
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(torch.bfloat16)


def gelu_t(x):
    return F.gelu(x.float()).to(torch.bfloat16)


def run(variant, ref):
    saved = {k: getattr(AG, k) for k in ("linear", "AttnFn", "LayerNormFn", "GeluFn")}
    try:
        if "lin" in variant:
            AG.linear = linear_t
        if "attn" in variant:
            AG.AttnFn = _Apply(attn_t)
        if "ln" in variant:
            AG.LayerNormFn = _Apply(ln_t)
        if "gelu" in variant:
            AG.GeluFn = _Apply(gelu_t)
        m = build()
        b = make_batch(CASES["11"], seed=4)
        out = m(**to_dev(b), inference=False)
        out["loss"].backward()
        got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    finally:
        for k, v in saved.items():
            setattr(AG, k, v)
    errs = {k: rl(got[k], r) for k, r in ref.items() if k in got and float(r.norm()) > 0}
    keys = ["text_hidden_fcs.0.0.weight", "text_hidden_fcs.0.2.weight", "grounding_encoder.sam2_model.sam_mask_decoder.output_hypernetworks_mlps.1.layers.0.weight",
            "grounding_encoder.sam2_model.sam_mask_decoder.output_hypernetworks_mlps.1.layers.2.weight",
            "grounding_encoder.sam2_model.sam_mask_decoder.transformer.layers.1.cross_attn_image_to_token.q_proj.weight",
            "grounding_encoder.sam2_model.sam_mask_decoder.transformer.layers.0.self_attn.q_proj.weight",
            "grounding_encoder.sam2_model.sam_mask_decoder.output_upscaling.0.weight", "grounding_encoder.sam2_model.sam_mask_decoder.mask_tokens.weight"]
    print(f"variant {sorted(variant)!s:40s} median {np.median(list(errs.values())):.4f} max {max(errs.values()):.4f} | " +
          " ".join(f"{errs.get(k, float('nan')):.3f}" for k in keys), flush=True)


def main():
    P, PS = params(G)
    for k in P:
        if ("text_hidden_fcs" in k) or k in ("lm_head.weight", "model.embed_tokens.weight"):
            P[k].requires_grad_(True)
    for k in PS:
        if k.startswith("sam_mask_decoder."):
            PS[k].requires_grad_(True)
    b = make_batch(CASES["11"], seed=4)
    U.model_forward(P, PS, oracle_cfg(), sam_cfg(), b, (1.0, 0.5, 2.0), SEG)["loss"].backward()
    ref = {k: v.grad for k, v in P.items() if v.requires_grad and v.grad is not None}
    ref.update({"grounding_encoder.sam2_model." + k: v.grad for k, v in PS.items() if v.requires_grad and v.grad is not None})
    print("columns: fcs.0.0.w fcs.0.2.w hyper1.l0.w hyper1.l2.w l1.i2t.q.w l0.self.q.w upscaling.0.w mask_tokens")
    for variant in (set(), {"lin"}, {"attn"}, {"ln"}, {"gelu"}, {"lin", "attn", "ln", "gelu"}):
        run(variant, ref)


if __name__ == "__main__":
    main()
