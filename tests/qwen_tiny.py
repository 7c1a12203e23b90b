"""Shared tiny-Qwen helpers for tests (config identical to tests/golden/make_qwen_fixtures.py TINY)."""
import os

import numpy as np
import torch

from oracle import qwen25vl as Q
from oracle.detweights import det_state_dict

GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "qwen_tiny.npz")


def gold():
    return np.load(GOLD_PATH, allow_pickle=False)


def oracle_cfg():
    return Q.QwenCfg(
        vision=Q.VisionCfg(depth=4, hidden_size=64, num_heads=4, intermediate_size=88, patch_size=14, temporal_patch_size=2,
                           spatial_merge_size=2, window_size=112, fullatt_block_indexes=(1, 3), out_hidden_size=96,
                           in_channels=3, tokens_per_second=2),
        text=Q.TextCfg(hidden_size=96, num_hidden_layers=2, num_attention_heads=6, num_key_value_heads=2, intermediate_size=160,
                       vocab_size=320, rms_norm_eps=1e-6, rope_theta=1000000.0, mrope_section=(2, 3, 3)),
        image_token_id=301, video_token_id=302, vision_start_token_id=303)


def product_cfg_kwargs():
    return dict(vocab_size=320, hidden_size=96, intermediate_size=160, num_hidden_layers=2, num_attention_heads=6,
                num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=1000000.0,
                rope_scaling={"type": "mrope", "mrope_section": [2, 3, 3]},
                vision_config=dict(depth=4, hidden_size=64, num_heads=4, intermediate_size=88, patch_size=14, temporal_patch_size=2,
                                   spatial_merge_size=2, window_size=112, fullatt_block_indexes=[1, 3], out_hidden_size=96,
                                   in_channels=3, tokens_per_second=2),
                image_token_id=301, video_token_id=302, vision_start_token_id=303, eos_token_id=319, pad_token_id=0)


def det_params(g, bf16_round=True):
    shapes = {str(n): eval(str(s)) for n, s in zip(g["param_names"], g["param_shapes"])}
    sd = det_state_dict(shapes, seed=1)
    if bf16_round:
        sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    return sd


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()
