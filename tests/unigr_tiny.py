"""Shared tiny-UniGR helpers: identical batch construction to tests/golden/make_unigr_fixtures.py."""
import os

import numpy as np
import torch

from oracle import sam2 as S
from oracle.detweights import det_state_dict, det_tensor
from tests.qwen_tiny import oracle_cfg, product_cfg_kwargs

SEG, T_SAM, SAM_SIDE = 300, 2, 1024
GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "unigr_tiny.npz")
SAM_TINY = dict(image_size=SAM_SIDE, embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4), pos_bkg=(7, 7),
                d_model=256, mem_dim=64, memattn_layers=2, memattn_ff=64)


def gold():
    return np.load(GOLD_PATH, allow_pickle=False)


def sam_cfg():
    return S.Sam2Cfg(image_size=SAM_SIDE, embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4), pos_bkg=(7, 7),
                     d_model=256, mem_dim=64, memattn_layers=2)


LABEL_HW = (120, 168)   # label / GT-mask resolution of the synthetic samples (non-square, not a divisor of 1024, like real data)


def params(g, bf16_round=False):
    """Name-derived deterministic weights; the SAM2 mask head's fitted read-out ("fit::<name>" in the fixture, tests/golden/blobfit.py) on top."""
    shapes = {str(n): eval(str(s)) for n, s in zip(g["param_names"], g["param_shapes"])}
    sam_shapes = {str(n): eval(str(s)) for n, s in zip(g["sam_param_names"], g["sam_param_shapes"])}
    P, PS = det_state_dict(shapes, seed=1), det_state_dict(sam_shapes, seed=2)
    for k in g.files:
        if k.startswith("fit::"):
            PS[k[5:]] = torch.from_numpy(g[k])
    # every weight of the fixture is bf16-representable (the generator pours the SAME rounded values into the reference's fp32 modules), so product,
    # oracle and reference compute on identical numbers; bf16_round is kept for callers of the old signature and changes nothing
    P = {k: v.to(torch.bfloat16).float() for k, v in P.items()}
    PS = {k: v.to(torch.bfloat16).float() for k, v in PS.items()}
    return P, PS


def make_batch(seg_flags, seed):
    g = np.random.default_rng(seed)
    grid = [[2, 8, 12]]
    nv = 2 * 4 * 6
    ids, labs = [], []
    for b, has in enumerate(seg_flags):
        pre = g.integers(0, 290, 6)
        ans = g.integers(0, 290, 7)
        if has:
            ans[3] = SEG
        seq = np.concatenate([pre, [303], np.full(nv, 302), g.integers(0, 290, 4), ans]).astype(np.int64)
        lab = np.full_like(seq, -100)
        lab[-7:] = seq[-7:]
        ids.append(seq); labs.append(lab)
    ids, labs = np.stack(ids), np.stack(labs)
    B = len(seg_flags)
    px = torch.cat([det_tensor(f"unigr_px_{seed}_{b}", (2 * 8 * 12, 1176), 1.0, seed=5) for b in range(B)], 0).to(torch.bfloat16).float()
    # SAM frames: one ellipse drifting over a smooth background per sample (tests/blob_inputs.py); GT masks = that object at label resolution
    from tests.blob_inputs import masks_at, object_video
    clips = [object_video(f"unigr_img_{seed}_{b}", T_SAM, SAM_SIDE, seed=6) for b in range(B)]
    imgs = torch.stack([c[0] for c in clips], 0).to(torch.bfloat16).float()   # inputs are bf16-representable too
    h, w = LABEL_HW
    masks = []
    for b, has in enumerate(seg_flags):
        m = masks_at(clips[b][1], (h, w))
        masks.append(m if has else m[0:0])
    return dict(input_ids=torch.from_numpy(ids), labels=torch.from_numpy(labs), attention_mask=torch.ones(B, ids.shape[1], dtype=torch.long),
                pixel_values_videos=px, video_grid_thw=torch.tensor(grid * B), second_per_grid_ts=torch.tensor([1.0] * B), images_sam=imgs,
                offset=torch.arange(B + 1), masks_list=masks, label_list=[torch.zeros(h, w) for _ in range(B)], resize_list=[(SAM_SIDE, SAM_SIDE)] * B)


def object_masks(seg_flags, seed):
    from tests.blob_inputs import object_video
    return [object_video(f"unigr_img_{seed}_{b}", T_SAM, SAM_SIDE, seed=6)[1] for b in range(len(seg_flags))]


CASES = {"11": (True, True), "10": (True, False), "00": (False, False)}
