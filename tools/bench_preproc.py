"""Input pipeline (SAM side and Qwen side): device time and HBM roofline of the resample / patchify passes.  python tools/bench_preproc.py"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.utils.preproc import sam_preprocess_frames
res = []
for (T, H, W) in [(16, 480, 854), (16, 720, 1280), (32, 1080, 1920)]:
    f = torch.randint(0, 256, (T, H, W, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        sam_preprocess_frames(f)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        sam_preprocess_frames(f)
    en.record(); en.synchronize()
    ms = st.elapsed_time(en) / 20
    alg = T * (H * W * 3 + H * 1024 * 3 * 2 + 1024 * 1024 * 3 * 2)   # read src, write+read the u8 row-resized image, write bf16 planes
    res.append({"T": T, "H": H, "W": W, "ms": round(ms, 4), "frames_per_s": round(T / ms * 1e3), "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 3)})
    print(res[-1])
# Qwen side: uint8 frames -> pixel_values_videos [N, 1176] bf16 (resize under max_pixels + normalise + patchify)
from rga3.utils.preproc import qwen_preprocess_video, smart_resize
qres = []
for (T, H, W, mp) in [(16, 480, 854, 384 * 784), (16, 720, 1280, 384 * 784), (32, 1080, 1920, 336 * 784), (16, 448, 448, 384 * 784)]:
    f = torch.randint(0, 256, (T, H, W, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        qwen_preprocess_video(f, max_pixels=mp)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        pv, g = qwen_preprocess_video(f, max_pixels=mp)
    en.record(); en.synchronize()
    ms = st.elapsed_time(en) / 20
    h, w = smart_resize(H, W, 28, 4 * 784, mp)
    alg = T * (H * W * 3 + (H * w * 3 * 2 if w != W else 0) + (h * w * 3 * 2 if (h, w) != (H, W) else h * w * 3)) + pv.numel() * 2
    qres.append({"T": T, "H": H, "W": W, "resized": [h, w], "rows": pv.shape[0], "ms": round(ms, 4), "frames_per_s": round(T / ms * 1e3),
                 "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 3)})
    print(qres[-1])
# CPU recipes (Pillow + torch) on this host, after all device timings (torch's CPU thread pool would disturb the launch loop)
from PIL import Image
x = np.random.randint(0, 256, (480, 854, 3), dtype=np.uint8)
t0 = time.perf_counter()
for _ in range(5):
    r = np.array(Image.fromarray(x, "RGB").resize((1024, 1024)))
    y = ((torch.from_numpy(r).permute(2, 0, 1).contiguous() - torch.tensor([123.675, 116.28, 103.53]).view(-1, 1, 1)) / torch.tensor([58.395, 57.12, 57.375]).view(-1, 1, 1)).bfloat16()
cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
print({"cpu_pillow_ms_per_frame_480x854": round(cpu_ms, 2)})
x = np.random.randint(0, 256, (16, 480, 854, 3), dtype=np.uint8)
t0 = time.perf_counter()
h, w = smart_resize(480, 854, 28, 4 * 784, 384 * 784)
r = np.stack([np.array(Image.fromarray(fr, "RGB").resize((w, h))) for fr in x])
v = torch.from_numpy(r).permute(0, 3, 1, 2).float()
v = (v - torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1) * 255) / (torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1) * 255)
pt = v.view(8, 2, 3, h // 28, 2, 14, w // 28, 2, 14).permute(0, 3, 6, 4, 7, 2, 1, 5, 8).reshape(-1, 1176).bfloat16()
qcpu_ms = (time.perf_counter() - t0) * 1e3
print({"cpu_recipe_ms_per_16_frame_clip_480x854": round(qcpu_ms, 2)})
json.dump({"gpu": res, "cpu_pillow_ms_per_frame_480x854": cpu_ms, "qwen_gpu": qres, "qwen_cpu_recipe_ms_per_16_frame_clip_480x854": qcpu_ms}, open(os.path.join(ROOT, "gpurun_out", "bench_preproc.json"), "w"), indent=1)
