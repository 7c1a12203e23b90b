"""Shared tiny-SAM2 helpers (config identical to tests/golden/make_sam2_fixtures.py TINY)."""
import os

import numpy as np
import torch

from oracle import sam2 as S
from oracle.detweights import det_state_dict, det_tensor

GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "sam2_tiny.npz")


def gold():
    return np.load(GOLD_PATH, allow_pickle=False)


def tiny_cfg():
    return S.Sam2Cfg(image_size=128, embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
                     pos_bkg=(7, 7), d_model=256, mem_dim=64, memattn_layers=2)


def det_params(g, bf16_round=False):
    shapes = {str(n): eval(str(s)) for n, s in zip(g["param_names"], g["param_shapes"])}
    sd = det_state_dict(shapes, seed=2)
    if bf16_round:
        sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    return sd


def images(T=5):
    return det_tensor("sam_images", (5, 3, 128, 128), 1.0, seed=3)[:T]


def lang(T=5):
    return det_tensor("lang_embd", (5, 1, 256), 1.0, seed=4)[:T]
