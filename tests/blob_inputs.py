"""Deterministic synthetic clips WITH AN OBJECT IN THEM (pure numpy; used by the fixture generators in tests/golden/, the oracle tests and the GPU tests).

White-noise frames through a random-weight SAM2 give speckle masks whose logits hover at zero — a sigmoid > 0.5 test on those pins nothing.
These clips show one coloured ellipse drifting over a low-frequency background; with the read-out of the mask head fitted to that object
(tests/golden/blobfit.py) the reference's masks are blobs with real logit margins, like a trained SAM2's."""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.default_rng((zlib.crc32(name.encode()) + 7919 * seed) & 0xFFFFFFFF)


def object_video(name: str, T: int, side: int, seed: int = 3, bg_amp: float = 0.4):
    """-> (frames fp32 [T, 3, side, side], object masks bool [T, side, side])."""
    rng = _rng(name, seed)
    lin = (np.arange(side, dtype=np.float32) + 0.5) / side
    y, x = np.meshgrid(lin, lin, indexing="ij")
    frames = np.zeros((T, 3, side, side), np.float32)
    for c in range(3):
        for _ in range(6):   # drifting plane waves, <= 2 cycles per image
            fx, fy = rng.uniform(-2, 2, 2)
            ph, a, dr = rng.uniform(0, 2 * np.pi), rng.uniform(0.5, 1.5), rng.uniform(-0.5, 0.5)
            for t in range(T):
                frames[t, c] += (bg_amp * a * np.sin(2 * np.pi * (fx * x + fy * y) + ph + dr * t)).astype(np.float32)
    cx0, cy0 = rng.uniform(0.4, 0.6, 2)
    vx, vy = rng.uniform(-0.04, 0.04, 2)
    rx, ry = rng.uniform(0.2, 0.3), rng.uniform(0.16, 0.24)
    colour = np.array([2.5, -2.0, 1.5], np.float32)   # every clip shows the same kind of object: the fitted read-out has to carry over to clips it never saw
    masks = np.zeros((T, side, side), bool)
    for t in range(T):
        masks[t] = ((x - cx0 - vx * t) / rx) ** 2 + ((y - cy0 - vy * t) / ry) ** 2 < 1.0
        frames[t] += masks[t][None] * colour[:, None, None]
    return torch.from_numpy(frames), torch.from_numpy(masks)


def masks_at(masks: torch.Tensor, hw):
    """Object masks resampled to a label size (area average > 0.5 when shrinking by an integer factor, nearest otherwise) -> float {0, 1} [T, h, w]."""
    m = masks.float()[:, None]
    H = m.shape[-1]
    if H % hw[0] == 0 and H % hw[1] == 0:
        m = torch.nn.functional.avg_pool2d(m, (H // hw[0], H // hw[1]))
    else:
        m = torch.nn.functional.interpolate(m, size=tuple(hw), mode="bilinear", align_corners=False)
    return (m[:, 0] > 0.5).float()
