"""Golden vectors for the SAM-side input pipeline, made in the survey container with the real Pillow (12.2.0).
The reference's utils/utils.py cannot be imported here (it needs torchvision / matplotlib), so its two functions are CALLED BY
RECIPE: DirectResize.apply_image = to_pil_image(image, 'RGB').resize((L, L)) (utils/utils.py:246-256) and
preprocess = (x - mean) / std (utils/utils.py:230-243) followed by .bfloat16() (inference_mevis.py:178-180).
    python tests/golden/make_preproc_fixtures.py"""
import os

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = {}
rng = np.random.default_rng(20260101)
sizes = [(120, 213), (270, 480), (300, 200)]
L = 256   # DirectResize(target_length); the reference uses 1024, the fixture a smaller target to stay small
mean = torch.tensor([123.675, 116.28, 103.53]).view(-1, 1, 1)
std = torch.tensor([58.395, 57.12, 57.375]).view(-1, 1, 1)
for i, (h, w) in enumerate(sizes):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
    img ^= rng.integers(0, 64, size=img.shape, dtype=np.uint8)
    res = np.array(Image.fromarray(img, "RGB").resize((L, L)))
    x = (torch.from_numpy(res).permute(2, 0, 1).contiguous() - mean) / std
    out[f"img{i}"] = img
    out[f"res{i}"] = res
    out[f"norm_bf16_sub{i}"] = x.bfloat16().float().numpy()[:, ::4, ::4]
out["n"] = np.int64(len(sizes))
out["L"] = np.int64(L)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "preproc.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
