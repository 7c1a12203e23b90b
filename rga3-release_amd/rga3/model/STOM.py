"""STOM — Spatio-Temporal Overlay Module, the numpy half (reference model/STOM.py:72-160).

The point tracker the reference delegates to (CoTracker3, third-party, not installed, weights absent) is a "next" row
(SURVEY.md 8(f).4): callers pass tracks in.  What is restated here is the reference's own arithmetic: median/MAD
filtering of the flow magnitudes, mean flow, integer-pixel shift of the RGBA visual prompt and alpha compositing.
"""
from __future__ import annotations

import numpy as np


def mean_flow(vip_track: np.ndarray, tgt_track: np.ndarray, visibility: np.ndarray):
    """reference STOM.py:102-130 -> (dx, dy) or None when the frame is left untouched."""
    vis = visibility.astype(bool)
    flows = tgt_track[vis] - vip_track[vis]
    if len(flows) == 0:
        return None
    mag = np.linalg.norm(flows, axis=1)
    med = np.median(mag)
    thr = 3 * np.median(np.abs(mag - med))
    keep = (mag >= med - thr) & (mag <= med + thr)
    f = flows[keep]
    if len(f) < visibility.shape[0] // 2:
        return None
    dx, dy = np.mean(f[:, 0]), np.mean(f[:, 1])
    if np.isnan(dx) or np.isnan(dy):
        return None
    return float(dx), float(dy)


def shift_overlay(src_rgba: np.ndarray, shape_hw, dx: float, dy: float) -> np.ndarray:
    """reference STOM.py:145-156: move every pixel with alpha > 0 by (int(x+dx), int(y+dy)) (truncation toward zero)."""
    out = np.zeros_like(src_rgba)
    ys, xs = np.nonzero(src_rgba[:, :, 3] > 0)
    nx = np.trunc(xs + dx).astype(np.int64)
    ny = np.trunc(ys + dy).astype(np.int64)
    ok = (nx >= 0) & (nx < shape_hw[1]) & (ny >= 0) & (ny < shape_hw[0])
    out[ny[ok], nx[ok]] = src_rgba[ys[ok], xs[ok]]
    return out


def composite(tgt_rgb: np.ndarray, overlay_rgba: np.ndarray) -> np.ndarray:
    from PIL import Image

    base = Image.fromarray(tgt_rgb, "RGB").convert("RGBA")
    return np.array(Image.alpha_composite(base, Image.fromarray(overlay_rgba, "RGBA")).convert("RGB"))


class STOM:
    def __init__(self, tracker=None):
        self.tracker = tracker  # callable(frames, src_vip, idx) -> (tracks [1,T,N,2], visibility [1,T,N]); CoTracker3 is not vendored

    def propagate_in_video(self, frames, src_frame_vip, vip_frame_idx, shape="rectangle", tracks=None, visibility=None):
        """frames: list of HxWx3 uint8; src_frame_vip: HxWx4 uint8 overlay -> list of HxWx3 uint8 (non-mask shapes)."""
        if tracks is None:
            if self.tracker is None:
                raise RuntimeError("STOM needs point tracks: no tracker is bundled (CoTracker3 is a third-party dependency of the reference)")
            tracks, visibility = self.tracker(frames, src_frame_vip, vip_frame_idx)
        if shape in ("mask", "mask contour"):
            raise NotImplementedError("mask-shaped prompts use cv2 morphology in the reference (STOM.py:163-207); cv2 is not available offline")
        out = []
        vip_track = tracks[0, vip_frame_idx]
        for i, f in enumerate(frames):
            if i == vip_frame_idx:
                out.append(composite(f, src_frame_vip))
                continue
            fl = mean_flow(vip_track, tracks[0, i], visibility[0, i])
            out.append(f if fl is None else composite(f, shift_overlay(src_frame_vip, f.shape[:2], fl[0], fl[1])))
        return out
