"""Golden vectors for the Qwen side of the input pipeline, made in the survey container with
  * the real Pillow (frame resize = PIL Image.resize, the call of qwen_vl_utils.fetch_image),
  * the installed transformers' own `smart_resize` and `Qwen2VLVideoProcessor.patchify`
    (models/qwen2_vl/video_processing_qwen2_vl.py:39-66, :236-274) called directly, and
  * the rescale + normalise step written with torch ops in both published orders (transformers 4.49 slow path / 5.x fused).
torchvision is not in the image and the HF module imports it at the top: an EMPTY module of that name is registered for the import
only -- neither function called here touches it.  qwen_vl_utils is absent: its list-of-frames recipe is called BY RECIPE
(smart_resize -> Image.resize -> pad to an even frame count), see oracle/preproc.py.
    python tests/golden/make_qwen_preproc_fixtures.py"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

from transformers.image_utils import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD   # before the stub: transformers must see torchvision as absent

for name in ["torchvision", "torchvision.transforms", "torchvision.transforms.v2", "torchvision.transforms.v2.functional"]:
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
from transformers.models.qwen2_vl.video_processing_qwen2_vl import Qwen2VLVideoProcessor, smart_resize  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rng = np.random.default_rng(20260102)
out = {}

# 1. smart_resize table: (h, w, min_pixels, max_pixels) -> (h', w')
rows = []
for _ in range(400):
    h, w = int(rng.integers(20, 2200)), int(rng.integers(20, 2200))
    mn = int(rng.choice([4 * 784, 3136, 100 * 784]))
    mx = int(rng.choice([336 * 784, 384 * 784, 1280 * 784, 16384 * 784]))
    if max(h, w) / min(h, w) > 200:
        continue
    rows.append((h, w, mn, mx) + tuple(smart_resize(h, w, 28, mn, mx)))
out["smart_resize"] = np.asarray(rows, np.int64)

# 2. whole recipe on small clips (odd frame count exercises the last-frame repeat)
cases = [(3, 75, 130, 6 * 784), (4, 100, 60, 8 * 784), (2, 56, 84, 16384 * 784)]   # (T, H, W, max_pixels)
mean = torch.tensor(OPENAI_CLIP_MEAN).view(1, 3, 1, 1)
std = torch.tensor(OPENAI_CLIP_STD).view(1, 3, 1, 1)
for i, (T, H, W, mx) in enumerate(cases):
    yy, xx = np.mgrid[0:H, 0:W]
    frames = np.stack([np.stack([(xx * 255 // (W - 1)), (yy * 255 // (H - 1)), ((xx + yy + 7 * t) % 256)], -1) for t in range(T)]).astype(np.uint8)
    frames ^= rng.integers(0, 64, size=frames.shape, dtype=np.uint8)
    h1, w1 = smart_resize(H, W, 28, 4 * 784, mx)                       # qwen_vl_utils.fetch_image
    res = np.stack([np.array(Image.fromarray(f, "RGB").resize((w1, h1))) for f in frames])
    h2, w2 = smart_resize(h1, w1, 28, 56 * 56, 12845056)               # HF processor's own bounds (Qwen2.5-VL preprocessor_config.json)
    print("case", i, (H, W), "->", (h1, w1), "->", (h2, w2))
    if (h2, w2) != (h1, w1):                                           # only when step 1 left the frame under 56*56 pixels
        res = np.stack([np.array(Image.fromarray(f, "RGB").resize((w2, h2))) for f in res])
    v = torch.from_numpy(res).permute(0, 3, 1, 2).contiguous()
    # transformers 4.49: rescale (float64 product -> float32) then (x - mean) / std
    x49 = torch.from_numpy((v.numpy().astype(np.float64) * (1 / 255)).astype(np.float32))
    n49 = (x49 - mean) / std
    # installed 5.x: fused mean/std (image_processing_backends.py:298-337)
    n5 = (v.float() - mean * (1.0 / (1 / 255))) / (std * (1.0 / (1 / 255)))
    p49, gt, gh, gw = Qwen2VLVideoProcessor.patchify(None, n49[None], 14, 2, 2)
    p5, *_ = Qwen2VLVideoProcessor.patchify(None, n5[None], 14, 2, 2)
    out[f"frames{i}"] = frames
    out[f"max_pixels{i}"] = np.int64(mx)
    out[f"res{i}"] = res
    out[f"pv49_{i}"] = p49[0].numpy()
    out[f"pv5_{i}"] = p5[0].numpy()
    out[f"grid{i}"] = np.asarray([gt, gh, gw], np.int64)
out["n"] = np.int64(len(cases))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "qwen_preproc.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()})
