"""SAM-side input pipeline on the GPU.  Host mirror of the reference's per-frame recipe

    image = DirectResize(L).apply_image(image_np)                       # utils/utils.py:246-256 (Pillow bicubic, uint8)
    image = preprocess(torch.from_numpy(image).permute(2, 0, 1))        # utils/utils.py:230-243 ((x - mean) / std)
    image = image.bfloat16()                                            # evaluation/mevis_val_u/inference_mevis.py:178-180

for a whole clip at once: uint8 frames [T, H, W, 3] already on the device -> images_sam [T, 3, L, L] bf16, bit-identical to the
CPU recipe.  The HIP extension is required (no CPU fallback)."""
from __future__ import annotations

import ctypes

import torch

from ..hip import lib as _lib

SAM_MEAN = (123.675, 116.28, 103.53)
SAM_STD = (58.395, 57.12, 57.375)
_tables = {}


def pil_bicubic_tables(in_size: int, out_size: int):
    """(bounds int32 [out, 2], kk int32 [out, ksize]) as CPU tensors -- Pillow's fixed-point bicubic tables (host-only C call)."""
    L = _lib.load()
    ks = ctypes.c_int(0)
    _lib.check(L.rga3_pil_bicubic_coeffs(in_size, out_size, None, None, 0, ctypes.addressof(ks)), "pil_bicubic_coeffs")
    bounds = torch.empty((out_size, 2), dtype=torch.int32)
    kk = torch.empty((out_size, ks.value), dtype=torch.int32)
    _lib.check(L.rga3_pil_bicubic_coeffs(in_size, out_size, bounds.data_ptr(), kk.data_ptr(), kk.numel(), ctypes.addressof(ks)), "pil_bicubic_coeffs")
    return bounds, kk


def _dev_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _tables:
        b, k = pil_bicubic_tables(in_size, out_size)
        _tables[key] = (b.to(device), k.to(device))
    return _tables[key]


def sam_preprocess_frames(frames_u8: torch.Tensor, size: int = 1024, mean=SAM_MEAN, std=SAM_STD, return_u8: bool = False):
    """frames_u8 [T, H, W, 3] uint8 (cuda, contiguous) -> bf16 [T, 3, size, size] (and the resized uint8 frames if asked)."""
    if not frames_u8.is_cuda:
        raise _lib.Rga3Error("sam_preprocess_frames needs device tensors (HIP path only)")
    assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3 and frames_u8.is_contiguous()
    T, H, W, _ = frames_u8.shape
    dev = frames_u8.device
    out = torch.empty((T, 3, size, size), dtype=torch.bfloat16, device=dev)
    u8 = torch.empty((T, size, size, 3), dtype=torch.uint8, device=dev) if return_u8 else None
    if H == size and W == size:   # Image.resize returns a copy: only the normalisation remains
        m = torch.tensor(mean, device=dev).view(1, 3, 1, 1)
        s = torch.tensor(std, device=dev).view(1, 3, 1, 1)
        out.copy_(((frames_u8.permute(0, 3, 1, 2).float() - m) / s))
        return (out, frames_u8.clone()) if return_u8 else out
    bh = kh = bv = kv = tmp = None
    ksh = ksv = 0
    if W != size:
        bh, kh = _dev_tables(W, size, dev)
        ksh = kh.shape[1]
        tmp = torch.empty((T, H, size, 3), dtype=torch.uint8, device=dev)
    if H != size:
        bv, kv = _dev_tables(H, size, dev)
        ksv = kv.shape[1]
    if H == size:   # horizontal pass only: resize to u8, normalise with torch ops (rare: frame height already 1024)
        u8h = torch.empty((T, size, size, 3), dtype=torch.uint8, device=dev)
        rc = _lib.load().rga3_sam_preprocess_u8(frames_u8.data_ptr(), T, H, W, size, size, bh.data_ptr(), kh.data_ptr(), ksh, None, None, 0,
                                                tmp.data_ptr(), u8h.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "sam_preprocess_u8")
        m = torch.tensor(mean, device=dev).view(1, 3, 1, 1)
        s = torch.tensor(std, device=dev).view(1, 3, 1, 1)
        out.copy_(((u8h.permute(0, 3, 1, 2).float() - m) / s))
        return (out, u8h) if return_u8 else out
    m3 = (ctypes.c_float * 3)(*mean)
    s3 = (ctypes.c_float * 3)(*std)
    p = lambda t: None if t is None else t.data_ptr()
    rc = _lib.load().rga3_sam_preprocess_u8(frames_u8.data_ptr(), T, H, W, size, size, p(bh), p(kh), ksh, p(bv), p(kv), ksv, p(tmp), p(u8), out.data_ptr(),
                                            ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "sam_preprocess_u8")
    return (out, u8) if return_u8 else out


# ------------------------------------------------------------------------------------------------ Qwen side of the row
# Host mirror of what the reference's callers do to a clip before UniGRModel sees it (evaluation/mevis_val_u/inference_mevis.py:196-216,
# utils/dataset.py:41-87):  process_vision_info(messages) -- qwen_vl_utils, list-of-frames branch: per frame smart_resize + PIL bicubic,
# pad to an even frame count -- then the HF processor's video branch: smart_resize with its own bounds, rescale 1/255, CLIP normalise,
# patchify to [N, 1176] (installed transformers models/qwen2_vl/video_processing_qwen2_vl.py:39-66, 236-336), then .bfloat16()
# (inference_mevis.py:214).  Here: uint8 frames on the device -> pixel_values_videos + video_grid_thw, in two or three launches.
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
QVU_MIN_PIXELS = 4 * 28 * 28
QVU_MAX_PIXELS = 16384 * 28 * 28
HF_MIN_PIXELS = 56 * 56
HF_MAX_PIXELS = 12845056
_luts = {}


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280):
    """Target (h, w): multiples of ``factor`` with min_pixels <= h*w <= max_pixels, aspect kept as closely as possible
    (video_processing_qwen2_vl.py:39-66; the same function in qwen_vl_utils)."""
    import math

    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
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


def qwen_norm_lut(device, mean=CLIP_MEAN, std=CLIP_STD, fused: bool = False) -> torch.Tensor:
    """[3, 256] fp32 on ``device``: the normalised value of every byte (host-only C call, cached)."""
    key = (str(device), tuple(mean), tuple(std), bool(fused))
    if key not in _luts:
        lut = torch.empty((3, 256), dtype=torch.float32)
        m3 = (ctypes.c_float * 3)(*mean)
        s3 = (ctypes.c_float * 3)(*std)
        _lib.check(_lib.load().rga3_qwen_norm_lut(ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), int(fused), lut.data_ptr()),
                   "qwen_norm_lut")
        _luts[key] = lut.to(device)
    return _luts[key]


def resize_frames_u8(frames_u8: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """PIL ``Image.resize((out_w, out_h))`` (bicubic) of every frame of uint8 [T, H, W, 3] on the device, bit-exact."""
    T, H, W, _ = frames_u8.shape
    if (H, W) == (out_h, out_w):
        return frames_u8
    dev = frames_u8.device
    out = torch.empty((T, out_h, out_w, 3), dtype=torch.uint8, device=dev)
    bh = kh = bv = kv = tmp = None
    ksh = ksv = 0
    if W != out_w:
        bh, kh = _dev_tables(W, out_w, dev)
        ksh = kh.shape[1]
        tmp = torch.empty((T, H, out_w, 3), dtype=torch.uint8, device=dev) if H != out_h else None
    if H != out_h:
        bv, kv = _dev_tables(H, out_h, dev)
        ksv = kv.shape[1]
    p = lambda t: None if t is None else t.data_ptr()
    if H == out_h:
        tmp = out   # horizontal pass only: it writes the result itself
    rc = _lib.load().rga3_sam_preprocess_u8(frames_u8.data_ptr(), T, H, W, out_h, out_w, p(bh), p(kh), ksh, p(bv), p(kv), ksv, p(tmp), out.data_ptr(), None,
                                            None, None, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "sam_preprocess_u8")
    return out


def qwen_preprocess_video(frames_u8: torch.Tensor, max_pixels: int = QVU_MAX_PIXELS, min_pixels: int = QVU_MIN_PIXELS,
                          hf_min_pixels: int = HF_MIN_PIXELS, hf_max_pixels: int = HF_MAX_PIXELS, out_dtype=torch.bfloat16, fused: bool = False,
                          patch: int = 14, tpatch: int = 2, merge: int = 2, mean=CLIP_MEAN, std=CLIP_STD, return_u8: bool = False):
    """frames_u8 [T, H, W, 3] uint8 (cuda, contiguous) -> (pixel_values_videos [N, 3*tpatch*patch*patch], video_grid_thw int64 [1, 3]).

    ``max_pixels`` / ``min_pixels`` are the per-frame bounds of the message element ({"type": "video", "max_pixels": ...}), ``hf_*`` the
    processor's own bounds.  ``fused`` picks the normalisation order of transformers 5.x instead of the reference's 4.49."""
    if not frames_u8.is_cuda:
        raise _lib.Rga3Error("qwen_preprocess_video needs device tensors (HIP path only)")
    assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3 and frames_u8.is_contiguous()
    T, H, W, _ = frames_u8.shape
    f = patch * merge
    h1, w1 = smart_resize(H, W, f, min_pixels, max_pixels)
    res = resize_frames_u8(frames_u8, h1, w1)
    h2, w2 = smart_resize(h1, w1, f, hf_min_pixels, hf_max_pixels)
    res = resize_frames_u8(res, h2, w2)
    gt, gh, gw = -(-T // tpatch), h2 // patch, w2 // patch
    out = torch.empty((gt * gh * gw, 3 * tpatch * patch * patch), dtype=out_dtype, device=frames_u8.device)
    assert out_dtype in (torch.bfloat16, torch.float32)
    lut = qwen_norm_lut(frames_u8.device, mean, std, fused)
    rc = _lib.load().rga3_qwen_patchify_u8(res.data_ptr(), T, h2, w2, lut.data_ptr(), out.data_ptr(), 0 if out_dtype == torch.bfloat16 else 1, patch, tpatch,
                                           merge, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "qwen_patchify_u8")
    grid = torch.tensor([[gt, gh, gw]], dtype=torch.int64, device=frames_u8.device)
    return (out, grid, res) if return_u8 else (out, grid)
