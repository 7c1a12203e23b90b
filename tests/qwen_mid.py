"""Shared helpers for the mid-size bf16 pin (tests/golden/qwen_mid_bf16.npz, made by tests/golden/make_qwen_mid_bf16_fixtures.py from the installed transformers):
the oracle's configuration, the weights regenerated from the stored shapes + seed (same rule as the generator), the inputs, and bf16-bit decoding."""
import os

import numpy as np
import torch

from oracle import qwen25vl as Q

GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "qwen_mid_bf16.npz")
VISION = dict(depth=8, hidden_size=256, num_heads=4, intermediate_size=688, patch_size=14, temporal_patch_size=2, spatial_merge_size=2, window_size=112,
              fullatt_block_indexes=(3, 7), out_hidden_size=512, in_channels=3, tokens_per_second=2)
TEXT = dict(hidden_size=512, num_hidden_layers=12, num_attention_heads=8, num_key_value_heads=2, intermediate_size=1408, vocab_size=2048, rms_norm_eps=1e-6,
            rope_theta=1000000.0, mrope_section=(8, 12, 12))
VIDEO_TOKEN = 2002


def gold():
    return np.load(GOLD_PATH, allow_pickle=False)


def oracle_cfg(layers=None):
    t = dict(TEXT)
    if layers is not None:
        t["num_hidden_layers"] = layers
    return Q.QwenCfg(vision=Q.VisionCfg(**VISION), text=Q.TextCfg(**t), image_token_id=2001, video_token_id=VIDEO_TOKEN, vision_start_token_id=2003)


def unbits(a):
    """uint16 bf16 bit patterns -> f32 tensor"""
    return torch.from_numpy(a.astype(np.int32) << 16).view(torch.float32).clone() if a.dtype == np.uint16 else torch.from_numpy(a)


def params(g):
    """The generator's mid_state_dict (sorted names, one generator, conditioned scales), rounded to bf16."""
    shapes = {str(n): eval(str(s)) for n, s in zip(g["param_names"], g["param_shapes"])}
    Lv, Lt = VISION["depth"], TEXT["num_hidden_layers"]
    gen = torch.Generator().manual_seed(int(g["seed"]))
    sd = {}
    for n in sorted(shapes):
        shp = tuple(shapes[n])
        if n == "model.embed_tokens.weight":
            t = torch.randn(shp, generator=gen)
        elif len(shp) >= 2:
            t = torch.randn(shp, generator=gen) * 0.02
            if n.startswith("visual.blocks.") and n.endswith(("attn.proj.weight", "mlp.down_proj.weight")):
                t = t * (2 * Lv) ** -0.5
            elif n.startswith("model.layers.") and n.endswith(("self_attn.o_proj.weight", "mlp.down_proj.weight")):
                t = t * (2 * Lt) ** -0.5
        elif "norm" in n or "ln_q" in n:
            t = 1.0 + 0.1 * torch.randn(shp, generator=gen)
        else:
            t = torch.randn(shp, generator=gen) * 0.02
        sd[n] = t.to(torch.bfloat16).float()
    return sd


def pixel_values(g):
    gen = torch.Generator().manual_seed(int(g["seed"]) + 1)
    t, h, w = (int(x) for x in g["grid"][0])
    return torch.randn(t * h * w, 1176, generator=gen).clamp_(-1.8, 2.2).to(torch.bfloat16).float()


def layer_params(P, k):
    """P with decoder layer k's tensors under the names of layer 0 (for a one-layer oracle run on a given input)."""
    pre = f"model.layers.{k}."
    out = {n: v for n, v in P.items() if not n.startswith("model.layers.")}
    out.update({"model.layers.0." + n[len(pre):]: v for n, v in P.items() if n.startswith(pre)})
    return out


def rel(a, b):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    return float((a - b).norm() / (b.norm() + 1e-12))
