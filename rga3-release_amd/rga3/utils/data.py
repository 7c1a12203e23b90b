"""Synthetic batch construction and the label-masking rule of the reference's collate_fn.

No tokenizer / processor / dataset exists offline, so batches are built from token-id arrays directly; the masking rule is
the pure function of reference utils/dataset.py:88-105.
"""
from __future__ import annotations

import numpy as np
import torch

from .staging import dict_to_cuda

IGNORE_INDEX = -100


def mask_labels(input_ids: np.ndarray, im_start_id: int, im_end_id: int, user_id: int, assistant_id: int, pad_token_id=None) -> np.ndarray:
    """labels = input_ids with everything except assistant spans set to -100.

    For every <|im_start|> ... <|im_end|> pair after the first (the system turn is skipped), an assistant turn keeps
    positions [start+3, end] (content tokens plus the closing <|im_end|>); user turns keep nothing; pad -> -100."""
    ids = np.asarray(input_ids)
    labels = ids.copy()
    keep = np.zeros(ids.shape, dtype=bool)
    for b in range(ids.shape[0]):
        starts = np.flatnonzero(ids[b] == im_start_id)
        ends = np.flatnonzero(ids[b] == im_end_id)
        for s, e in zip(starts[1:], ends[1:]):
            if s + 1 < ids.shape[1] and ids[b, s + 1] == assistant_id:
                keep[b, s + 3: e + 1] = True
    labels[~keep] = IGNORE_INDEX
    if pad_token_id is not None:
        labels[labels == pad_token_id] = IGNORE_INDEX
    return labels


def make_batch(cfg, device, batch=1, frames_mllm=16, frames_sam=16, side=448, sam_side=1024, n_text=64, seed=0, seg=True, label_hw=(480, 640),
               dtype=torch.bfloat16, seg_pos=-3, ints_only=False):
    """Synthetic UniGRModel kwargs of the shape the reference's collate_fn produces:
    frames_mllm frames side x side -> video_grid_thw [[frames/2, side/14, side/14]], n_text non-video tokens incl. the
    answer "Sure, [SEG]." (6 supervised tokens, [SEG] at `seg_pos` in [-5, -2]), random rectangle GT masks.  The batch is built on the CPU, as the
    reference's collate_fn builds it, and moved by dict_to_cuda (device=None: the CPU dict is returned; ints_only: only the integer tensors -- what
    changes from step to step when a benchmark keeps a pool of pixel tensors resident in HBM)."""
    g = torch.Generator().manual_seed(seed)
    gt, gh = frames_mllm // 2, side // 14
    n_vid = gt * (gh // 2) * (gh // 2)
    ids, labels = [], []
    for b in range(batch):
        text = torch.randint(1000, 100000, (n_text,), generator=g)
        seq = torch.cat([text[:10], torch.tensor([cfg.vision_start_token_id]), torch.full((n_vid,), cfg.video_token_id),
                         torch.tensor([cfg.vision_end_token_id]), text[12:]])
        lab = torch.full_like(seq, IGNORE_INDEX)
        if seg:
            seq[seg_pos] = cfg.seg_token_idx
        lab[-6:] = seq[-6:]
        ids.append(seq)
        labels.append(lab)
    input_ids, labels = torch.stack(ids), torch.stack(labels)
    ints = dict(input_ids=input_ids, attention_mask=torch.ones_like(input_ids), labels=labels)
    if ints_only:
        return ints if device is None else dict_to_cuda(ints, device)
    px = torch.randn(batch * gt * gh * gh, 1176, generator=g).clamp_(-1.8, 2.2).to(dtype)
    images_sam = torch.randn(batch, frames_sam, 3, sam_side, sam_side, generator=g).to(dtype)
    masks = []
    for b in range(batch):
        m = torch.zeros(frames_sam if seg else 0, *label_hw)
        for t in range(m.shape[0]):
            y0, x0 = int(torch.randint(0, label_hw[0] // 2, (1,), generator=g)), int(torch.randint(0, label_hw[1] // 2, (1,), generator=g))
            m[t, y0:y0 + label_hw[0] // 3, x0:x0 + label_hw[1] // 3] = 1
        masks.append(m)
    label_list = [torch.zeros(label_hw) for _ in range(batch)]     # only its shape is read (reference qwen_2_5_vl_sam2.py:268): stays on the host
    d = dict(ints, pixel_values_videos=px, video_grid_thw=torch.tensor([[gt, gh, gh]] * batch), second_per_grid_ts=torch.ones(batch), images_sam=images_sam,
             offset=torch.arange(batch + 1), masks_list=masks)
    if device is not None:
        d = dict_to_cuda(d, device)
    return dict(d, label_list=label_list, resize_list=[(sam_side, sam_side)] * batch, inference=False)


# ------------------------------------------------------------------------------------------------ frame selection (host side)
# What the reference's inference scripts do before any preprocessing (evaluation/mevis_val_u/inference_mevis.py:156-160 and the other
# inference_*.py): choose num_frames_mllm frame indices of the clip for the MLLM, and num_frames_sam of those for SAM2
# (reference utils/utils.py:201-229).  Pinned by tests/golden/frame_sampling.npz (made by executing the reference's functions).
def _bin_centres(lo: int, hi: int, n: int):
    """Centres (floor) of the n integer bins [e_i, e_{i+1} - 1] whose edges are the truncated np.linspace(lo, hi, n + 1)."""
    edges = np.linspace(start=lo, stop=hi, num=n + 1).astype(int)
    return ((edges[:-1] + edges[1:] - 1) // 2).tolist()


def uniform_sample(total_len: int, sample_num: int):
    """sample_num frame indices spread evenly over [0, total_len) (utils/utils.py:201-208)."""
    return _bin_centres(0, total_len, sample_num)


def get_sparse_indices(total_frame_num: int, num_frames_mllm: int):
    """Sorted indices of the frames shown to the MLLM: even sampling of a long clip; a short clip is repeated whole
    num_frames_mllm // total times plus an even sample of the remainder (utils/utils.py:211-219)."""
    if total_frame_num > num_frames_mllm:
        return sorted(uniform_sample(total_frame_num, num_frames_mllm))
    rep, extra = divmod(num_frames_mllm, total_frame_num)
    return sorted(list(range(total_frame_num)) * rep + uniform_sample(total_frame_num, extra))


def get_dense_indices(num_frames_mllm: int, num_frames_sam: int):
    """Positions (within the MLLM frame list) of the frames SAM2 segments (utils/utils.py:222-229: bins over [0, num_frames_mllm - 1])."""
    return _bin_centres(0, num_frames_mllm - 1, num_frames_sam)
