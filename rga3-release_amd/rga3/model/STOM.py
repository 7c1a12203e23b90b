"""STOM — Spatio-Temporal Overlay Module, the numpy half (reference model/STOM.py:72-207).

The point tracker the reference delegates to (CoTracker3, third-party, not installed, weights absent) is a "next" row
(SURVEY.md 8(f).4): callers pass tracks in.  What is restated here is the reference's own arithmetic: median/MAD
filtering of the flow magnitudes, mean flow, integer-pixel shift of the RGBA visual prompt and alpha compositing
(shapes other than masks, :102-160), and for mask-shaped prompts the tracked-point raster -> morphological closing ->
centroid -> filled circle of :163-207.  The reference calls OpenCV for the last four (cv2.getStructuringElement / morphologyEx /
moments / circle); cv2 is not in this image, so those are restated from OpenCV 4.x's published algorithms (imgproc morph.cpp,
drawing.cpp) -- parity with the installed cv2 of a reference deployment is unpinned, the structuring elements are checked against
OpenCV's documented 3x3 / 5x5 ellipses.
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


def ellipse_kernel(k: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)) (OpenCV morph.cpp): row i spans c - dx .. c + dx with
    dx = round(c * sqrt((r^2 - (i - r)^2) / r^2)), r = c = k // 2."""
    if k <= 1:
        return np.ones((max(k, 1), max(k, 1)), np.uint8)
    r = c = k // 2
    out = np.zeros((k, k), np.uint8)
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) / float(r * r))))
            out[i, max(c - dx, 0): min(c + dx + 1, k)] = 1
    return out


def _morph(img: np.ndarray, kernel: np.ndarray, dilate: bool) -> np.ndarray:
    """cv2.dilate / cv2.erode with the anchor at the kernel centre and the default border (outside pixels never win):
    dst(y, x) = max / min over kernel elements (i, j) != 0 of src(y + i - ay, x + j - ax)."""
    h, w = img.shape
    ay, ax = kernel.shape[0] // 2, kernel.shape[1] // 2
    out = np.zeros_like(img) if dilate else np.full_like(img, 255)
    for i, j in np.argwhere(kernel > 0):
        dy, dx = i - ay, j - ax
        ys0, ys1 = max(0, -dy), min(h, h - dy)        # destination rows whose source row y + dy is inside
        xs0, xs1 = max(0, -dx), min(w, w - dx)
        if ys0 >= ys1 or xs0 >= xs1:
            continue
        src = img[ys0 + dy: ys1 + dy, xs0 + dx: xs1 + dx]
        dst = out[ys0:ys1, xs0:xs1]
        np.maximum(dst, src, out=dst) if dilate else np.minimum(dst, src, out=dst)
    return out


def morph_close(mask: np.ndarray, k: int) -> np.ndarray:
    """cv2.morphologyEx(mask, cv2.MORPH_CLOSE, ellipse(k)): dilation, then erosion, same kernel and anchor."""
    ker = ellipse_kernel(k)
    return _morph(_morph(mask, ker, True), ker, False)


def filled_circle(shape_hw, cx: int, cy: int, radius: int) -> np.ndarray:
    """cv2.circle(img, (cx, cy), radius, 255, -1) on a zero uint8 image (OpenCV drawing.cpp Circle(): midpoint algorithm, horizontal spans)."""
    h, w = shape_hw
    img = np.zeros((h, w), np.uint8)

    def hline(y, x0, x1):
        if 0 <= y < h:
            x0, x1 = max(x0, 0), min(x1, w - 1)
            if x0 <= x1:
                img[y, x0:x1 + 1] = 255

    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        hline(cy - dy, cx - dx, cx + dx)
        hline(cy + dy, cx - dx, cx + dx)
        hline(cy - dx, cx - dy, cx + dy)
        hline(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        m = -1 if err > 0 else 0          # (err <= 0) - 1
        err -= minus & m
        dx += m
        minus -= m & 2
    return img


def warp_point(src_rgba: np.ndarray, tgt_rgb: np.ndarray, tracks: np.ndarray, visibility: np.ndarray):
    """reference STOM.py:163-207 (mask-shaped prompts): the visible tracked points are rasterised, closed with an elliptical kernel of min(h, w) // 15,
    and a filled circle of radius min(h, w) // 20 in the prompt's colour (alpha clamped to [96, 148]) is drawn at the centroid of the closed mask.
    Returns the composited RGB frame (the frame itself when fewer than half of the points are visible)."""
    vis = visibility.astype(bool)
    if vis.sum() < len(tracks) // 2:
        return tgt_rgb
    on = src_rgba[:, :, 3] > 0
    colour = src_rgba[on][0].copy() if on.any() else np.zeros(4, np.uint8)
    colour[3] = max(min(int(colour[3]), 148), 96)
    h, w = src_rgba.shape[:2]
    mask = np.zeros((h, w), np.uint8)
    for pt, v in zip(tracks, vis):
        if v:
            x, y = int(pt[1]), int(pt[0])          # the reference's naming: x is the ROW, y the column
            if 0 <= x < h and 0 <= y < w:
                mask[x, y] = 255
    closed = morph_close(mask, min(h, w) // 15)
    overlay = np.zeros_like(src_rgba)
    m00 = float(closed.astype(np.float64).sum())
    if m00 != 0:
        ys, xs = np.nonzero(closed)
        wgt = closed[ys, xs].astype(np.float64)
        cx, cy = int((xs * wgt).sum() / m00), int((ys * wgt).sum() / m00)
        overlay[filled_circle((h, w), cx, cy, min(h, w) // 20) > 0] = colour
    return composite(tgt_rgb, overlay)


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
        out = []
        vip_track = tracks[0, vip_frame_idx]
        for i, f in enumerate(frames):
            if i == vip_frame_idx:
                out.append(composite(f, src_frame_vip))
                continue
            if shape in ("mask", "mask contour"):
                try:
                    out.append(warp_point(src_frame_vip, f, tracks[0, i], visibility[0, i]))
                except Exception:      # the reference swallows every failure of this branch and keeps the frame (:95-101)
                    out.append(f)
                continue
            fl = mean_flow(vip_track, tracks[0, i], visibility[0, i])
            out.append(f if fl is None else composite(f, shift_overlay(src_frame_vip, f.shape[:2], fl[0], fl[1])))
        return out
