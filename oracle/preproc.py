"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's SAM-side input pipeline:

    image  = DirectResize(1024).apply_image(image_np)      reference utils/utils.py:246-256  (PIL Image.resize, default BICUBIC)
    image  = preprocess(torch.from_numpy(image).permute(2, 0, 1))   utils/utils.py:230-243   ((x - mean) / std, fp32)
    image  = image.bfloat16()                              evaluation/mevis_val_u/inference_mevis.py:178-180

The resize is a third-party dependency (Pillow; 12.2.0 in this image, un-vendored): its published algorithm (src/libImaging/
Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) is restated here in
numpy with Python floats (IEEE double, same operation order as the C) and PINNED bit-exactly against Pillow itself in
tests/test_oracle_preproc.py and against tests/golden/preproc.npz.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import this module."""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2   # Resample.c: 8 bits for the result, 2 bits of headroom for the accumulator
SAM_MEAN = (123.675, 116.28, 103.53)   # utils/utils.py:232
SAM_STD = (58.395, 57.12, 57.375)      # utils/utils.py:233


def _bicubic(x: float) -> float:
    """Resample.c bicubic_filter, a = -0.5, support 2."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the full box (in0 = 0, in1 = in_size).
    Returns (bounds int32 [out, 2] = (first source index, tap count), kk int32 [out, ksize], ksize)."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)      # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(src: np.ndarray, bounds: np.ndarray, kk: np.ndarray) -> np.ndarray:
    """One 8-bit resampling pass along axis 0 of src [n_in, ...] -> [n_out, ...] (int32 accumulate, round, >> 22, clip)."""
    n_out = bounds.shape[0]
    out = np.empty((n_out,) + src.shape[1:], np.uint8)
    s32 = src.astype(np.int64)
    for i in range(n_out):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.tensordot(kk[i, :n].astype(np.int64), s32[lo:lo + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL.Image.fromarray(img, 'RGB').resize((out_w, out_h)) for uint8 [H, W, 3]: horizontal pass over the rows the vertical pass
    will use, then the vertical pass (ImagingResample: two passes, uint8 intermediate)."""
    H, W, _ = img.shape
    if (H, W) == (out_h, out_w):
        return img.copy()
    bh, kh, _ = pil_bicubic_coeffs(W, out_w)
    bv, kv, _ = pil_bicubic_coeffs(H, out_h)
    cur = img
    if W != out_w:
        first = int(bv[0, 0])
        last = int(bv[-1, 0] + bv[-1, 1])
        if H != out_h:
            cur = cur[first:last]
            bv = bv.copy()
            bv[:, 0] -= first
        cur = _pass(np.ascontiguousarray(cur.transpose(1, 0, 2)), bh, kh).transpose(1, 0, 2)
    if H != out_h:
        cur = _pass(np.ascontiguousarray(cur), bv, kv)
    return np.ascontiguousarray(cur)


def sam_preprocess(frames_u8: np.ndarray, size: int = 1024):
    """[T, H, W, 3] uint8 -> (resized uint8 [T, size, size, 3], normalised fp32 [T, 3, size, size]); the reference then casts to bf16."""
    import torch

    res = np.stack([resize_bicubic_u8(f, size, size) for f in frames_u8])
    x = torch.from_numpy(res).permute(0, 3, 1, 2).contiguous()
    mean = torch.tensor(SAM_MEAN).view(-1, 1, 1)
    std = torch.tensor(SAM_STD).view(-1, 1, 1)
    return res, (x - mean) / std
