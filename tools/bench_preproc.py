"""SAM-side input pipeline: device time and HBM roofline of the two resample passes.  python tools/bench_preproc.py"""
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
# CPU recipe (Pillow + torch) on this host for one frame
from PIL import Image
x = np.random.randint(0, 256, (480, 854, 3), dtype=np.uint8)
t0 = time.perf_counter()
for _ in range(5):
    r = np.array(Image.fromarray(x, "RGB").resize((1024, 1024)))
    y = ((torch.from_numpy(r).permute(2, 0, 1).contiguous() - torch.tensor([123.675, 116.28, 103.53]).view(-1, 1, 1)) / torch.tensor([58.395, 57.12, 57.375]).view(-1, 1, 1)).bfloat16()
cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
print({"cpu_pillow_ms_per_frame_480x854": round(cpu_ms, 2)})
json.dump({"gpu": res, "cpu_pillow_ms_per_frame_480x854": cpu_ms}, open(os.path.join(ROOT, "gpurun_out", "bench_preproc.json"), "w"), indent=1)
