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
