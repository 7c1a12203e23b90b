"""Mask metrics of the reference's validation / evaluation harness, vectorised.

 * intersection_and_union  — utils/utils.py:140-152 (histc over K classes, ignore_index copied into the prediction)
 * giou_ciou               — train_joint.py:615-641 (gIoU = mean per-frame IoU with empty-target = 1; cIoU = sum I / sum U)
 * db_eval_iou             — evaluation/mevis_val_u/metrics.py:6-37 (the "mask IoU" of the parity metric)
"""
from __future__ import annotations

import numpy as np
import torch


def intersection_and_union(output: torch.Tensor, target: torch.Tensor, K: int, ignore_index: int = 255):
    assert output.dim() in (1, 2, 3) and output.shape == target.shape
    o = output.reshape(-1).clone()
    t = target.reshape(-1)
    o[t == ignore_index] = ignore_index
    inter = o[o == t]
    cnt = lambda v: torch.bincount(v[(v >= 0) & (v < K)].long(), minlength=K).float()
    ai, ao, at = cnt(inter), cnt(o), cnt(t)
    return ai, ao + at - ai, at


class GIoUCIoU:
    """Accumulator reproducing train_joint.py:586-648 for one rank (call all_reduce_sums across ranks if distributed)."""

    def __init__(self):
        self.inter = np.zeros(2)
        self.union = np.zeros(2)
        self.acc = np.zeros(2)
        self.count = 0

    def update(self, pred_masks: torch.Tensor, gt_masks: torch.Tensor):
        """pred/gt [T, h, w] (bool or int)."""
        inter, union, acc = torch.zeros(2), torch.zeros(2), torch.zeros(2)
        for m, o in zip(gt_masks.int(), pred_masks.int()):
            i, u, _ = intersection_and_union(o.contiguous().clone(), m.contiguous(), 2, ignore_index=255)
            i, u = i.cpu(), u.cpu()
            inter += i
            union += u
            a = i / (u + 1e-5)
            a[u == 0] += 1.0  # no-object target
            acc += a
        n = gt_masks.shape[0]
        self.inter += inter.numpy()
        self.union += union.numpy()
        self.acc += acc.numpy() / n * n
        self.count += n

    def sums(self):
        return np.concatenate([self.inter, self.union, self.acc, [self.count]])

    def load_sums(self, s):
        self.inter, self.union, self.acc, self.count = s[0:2], s[2:4], s[4:6], float(s[6])

    def compute(self):
        iou_class = self.inter / (self.union + 1e-10)
        return float(self.acc[1] / max(self.count, 1e-5)), float(iou_class[1])  # (giou, ciou)


def db_eval_iou(annotation: np.ndarray, segmentation: np.ndarray, void_pixels=None):
    assert annotation.shape == segmentation.shape
    a, s = annotation.astype(bool), segmentation.astype(bool)
    v = np.zeros_like(s) if void_pixels is None else void_pixels.astype(bool)
    inters = np.sum((s & a) & ~v, axis=(-2, -1))
    union = np.sum((s | a) & ~v, axis=(-2, -1))
    with np.errstate(divide="ignore", invalid="ignore"):
        j = inters / union
    if np.ndim(j) == 0:
        return 1 if np.isclose(union, 0) else j
    j[np.isclose(union, 0)] = 1
    return j
