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
    """Name-derived deterministic weights + the fitted mask-head read-out stored in the fixture ("fit::<name>", tests/golden/blobfit.py)."""
    shapes = {str(n): eval(str(s)) for n, s in zip(g["param_names"], g["param_shapes"])}
    sd = det_state_dict(shapes, seed=2)
    for k in g.files:
        if k.startswith("fit::"):
            sd[k[5:]] = torch.from_numpy(g[k])
    # every weight of the fixture is bf16-representable (the generator pours the SAME rounded values into the reference's fp32 modules): product (bf16
    # storage), oracle and reference compute on identical numbers, so comparisons against the reference's own outputs measure arithmetic only
    sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    if bf16_round:
        sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    return sd


def images(T=5):
    """The fixture clip: an ellipse drifting over a smooth background (tests/blob_inputs.py)."""
    from tests.blob_inputs import object_video
    return object_video("sam_images", 5, 128, seed=3)[0][:T].to(torch.bfloat16).float()


def object_masks(T=5):
    from tests.blob_inputs import object_video
    return object_video("sam_images", 5, 128, seed=3)[1][:T]


def lang(T=5):
    return det_tensor("lang_embd", (5, 1, 256), 1.0, seed=4)[:T].to(torch.bfloat16).float()


def lang2(T=5):
    """The second object's prompts of tests/golden/sam2_multiobj.npz (make_sam2_multiobj_fixtures.second_prompt)."""
    return det_tensor("lang_embd_obj1", (5, 1, 256), 4.0, seed=9)[:T].to(torch.bfloat16).float()


def gold_multiobj():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "sam2_multiobj.npz"), allow_pickle=False)
