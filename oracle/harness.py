"""ORACLE — test infrastructure only.  Loop restatements of the reference's harness-side integer logic:
 * label masking        — reference utils/dataset.py:88-105
 * intersectionAndUnion — reference utils/utils.py:140-152 (histc form)
 * gIoU / cIoU          — reference train_joint.py:615-641
 * STOM warp / MAD      — reference model/STOM.py:102-156
"""
import numpy as np
import torch


def mask_labels_ref(input_ids, im_start, im_end, user, assistant, pad=None):
    ids = torch.as_tensor(np.asarray(input_ids))
    labels = ids.clone()
    masks = torch.ones_like(labels).bool()
    for b in range(ids.shape[0]):
        s_all = torch.where(ids[b] == im_start)[0]
        e_all = torch.where(ids[b] == im_end)[0]
        for start, end in zip(s_all[1:], e_all[1:]):
            if ids[b][start + 1] == user:
                continue
            elif ids[b][start + 1] == assistant:
                masks[b][start + 3: end + 1] = False
    labels[masks] = -100
    if pad is not None:
        labels[labels == pad] = -100
    return labels.numpy()


def intersection_and_union_ref(output, target, K, ignore_index=255):
    output = output.reshape(-1).clone().float()
    target = target.reshape(-1).float()
    output[target == ignore_index] = ignore_index
    inter = output[output == target]
    ai = torch.histc(inter, bins=K, min=0, max=K - 1)
    ao = torch.histc(output, bins=K, min=0, max=K - 1)
    at = torch.histc(target, bins=K, min=0, max=K - 1)
    return ai, ao + at - ai, at


def giou_ciou_ref(pairs):
    """pairs: list of (pred [T,h,w] int, gt [T,h,w] int) — one validation sample each."""
    I = np.zeros(2); U = np.zeros(2); A = np.zeros(2); n_tot = 0
    for pred, gt in pairs:
        inter, union, acc = 0.0, 0.0, 0.0
        for m, o in zip(gt.int(), pred.int()):
            i, u, _ = intersection_and_union_ref(o.contiguous().clone(), m.contiguous(), 2)
            inter += i; union += u
            acc += i / (u + 1e-5)
            acc[u == 0] += 1.0
        n = gt.shape[0]
        I += inter.numpy(); U += union.numpy(); A += acc.numpy() / n * n; n_tot += n
    return A[1] / n_tot, (I / (U + 1e-10))[1]


def stom_shift_ref(src, shape_hw, fx, fy):
    out = np.zeros_like(src)
    for y, x in np.argwhere(src[:, :, 3] > 0):
        nx, ny = int(x + fx), int(y + fy)
        if 0 <= nx < shape_hw[1] and 0 <= ny < shape_hw[0]:
            out[ny, nx, :] = src[y, x, :]
    return out
