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


# ------------------------------------------------------------------------------------------------ Qwen side of the row
# Reference call path (evaluation/mevis_val_u/inference_mevis.py:196-216, same in the other inference scripts; training:
# utils/dataset.py:41-87): messages carry {"type": "video", "video": [PIL frames], "max_pixels": P};
#   1. qwen_vl_utils.process_vision_info (third party, pinned ==0.0.8 by the reference's requirements, ABSENT from this image),
#      list-of-frames branch of fetch_video: every frame -> fetch_image: convert("RGB"), smart_resize(h, w, factor=28,
#      min_pixels = ele.get("min_pixels", 4*28*28), max_pixels = ele.get("max_pixels", 16384*28*28)), PIL image.resize((w', h'))
#      (default filter BICUBIC); the list is padded to an even length by repeating the last frame.  Restated from the published
#      source -- this glue is PARITY UNPINNED; its two arithmetic pieces are pinned: smart_resize against the identical function in
#      the installed transformers (models/qwen2_vl/video_processing_qwen2_vl.py:39-66) and the resize against Pillow itself.
#   2. the HF processor: smart_resize again with the processor's own bounds (identity for frames already sized in step 1 unless
#      they fall outside them), rescale 1/255, CLIP normalise, patchify (video_processing_qwen2_vl.py:236-336).  patchify is pinned
#      against the installed method; the normalise order has two published forms (4.49 slow path / 5.x fused), both restated.
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # transformers image_utils.OPENAI_CLIP_MEAN
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)   # transformers image_utils.OPENAI_CLIP_STD
QVU_MIN_PIXELS = 4 * 28 * 28          # qwen_vl_utils.vision_process.MIN_PIXELS
QVU_MAX_PIXELS = 16384 * 28 * 28      # qwen_vl_utils.vision_process.MAX_PIXELS
HF_MIN_PIXELS = 56 * 56               # Qwen2.5-VL preprocessor_config.json min_pixels (3136)
HF_MAX_PIXELS = 12845056              # Qwen2.5-VL preprocessor_config.json max_pixels


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280):
    """video_processing_qwen2_vl.py:39-66 (== qwen_vl_utils.smart_resize): both sides multiples of factor, area within bounds."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def qwen_norm_lut(mean=CLIP_MEAN, std=CLIP_STD, fused: bool = False) -> np.ndarray:
    """[3, 256] fp32: the normalised value of every byte per channel.
    fused=False: transformers 4.49 image_transforms.rescale (float64 product cast to float32) then normalize ((x - mean) / std, float32);
    fused=True : installed 5.x image_processing_backends.py:298-337 (mean, std pre-multiplied by 255 in float32)."""
    b = np.arange(256)
    m = np.asarray(mean, np.float32)[:, None]
    s = np.asarray(std, np.float32)[:, None]
    if fused:
        m255 = (m * np.float32(1.0 / (1 / 255)))
        s255 = (s * np.float32(1.0 / (1 / 255)))
        return ((b.astype(np.float32)[None] - m255) / s255).astype(np.float32)
    x = (b.astype(np.float64) * (1 / 255)).astype(np.float32)[None]
    return ((x - m) / s).astype(np.float32)


def qwen_patchify(video: np.ndarray, patch: int = 14, tpatch: int = 2, merge: int = 2):
    """video [T, C, h, w] -> ([gt*gh*gw, C*tpatch*patch*patch], (gt, gh, gw)); video_processing_qwen2_vl.py:236-274."""
    T, C, h, w = video.shape
    if T % tpatch:
        video = np.concatenate([video, np.repeat(video[-1:], tpatch - T % tpatch, 0)], 0)
        T = video.shape[0]
    gt, gh, gw = T // tpatch, h // patch, w // patch
    x = video.reshape(gt, tpatch, C, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.transpose(0, 3, 6, 4, 7, 2, 1, 5, 8)
    return np.ascontiguousarray(x).reshape(gt * gh * gw, C * tpatch * patch * patch), (gt, gh, gw)


def qwen_video_preprocess(frames_u8, min_pixels: int = QVU_MIN_PIXELS, max_pixels: int = QVU_MAX_PIXELS, hf_min_pixels: int = HF_MIN_PIXELS,
                          hf_max_pixels: int = HF_MAX_PIXELS, fused: bool = False, patch: int = 14, tpatch: int = 2, merge: int = 2):
    """frames uint8 [T, H, W, 3] -> (pixel_values_videos fp32 [N, 1176], grid (t, h, w), resized uint8 [T, h', w', 3])."""
    T, H, W, _ = frames_u8.shape
    f = patch * merge
    h1, w1 = smart_resize(H, W, f, min_pixels, max_pixels)
    res = np.stack([resize_bicubic_u8(fr, h1, w1) for fr in frames_u8])
    h2, w2 = smart_resize(h1, w1, f, hf_min_pixels, hf_max_pixels)
    if (h2, w2) != (h1, w1):
        res = np.stack([resize_bicubic_u8(fr, h2, w2) for fr in res])
    lut = qwen_norm_lut(fused=fused)
    chw = res.transpose(0, 3, 1, 2)
    norm = np.stack([lut[c][chw[:, c]] for c in range(3)], 1)
    pv, grid = qwen_patchify(norm, patch, tpatch, merge)
    return pv, grid, res
