"""Would running the SAM2-L encoder beside the Qwen forward (two streams) shorten evaluate()?  python tools/overlap_probe.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import bench
from rga3.model.qwen2_5_vl import Qwen2_5_VLForConditionalGeneration
dev = torch.device("cuda:0")
model, cfg, b = bench.build_full(dev, 0, 16)
model.eval()
gm = model.grounding_encoder
kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], pixel_values_videos=b["pixel_values_videos"], video_grid_thw=b["video_grid_thw"],
          second_per_grid_ts=b.get("second_per_grid_ts"), output_hidden_states=True)
def qwen():
    return Qwen2_5_VLForConditionalGeneration.forward(model, **kw)
def enc():
    return gm.get_sam2_embeddings(b["images_sam"][0])._ensure_feats()
side = torch.cuda.Stream()
def seq():
    qwen(); enc()
def par():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        f = enc()
    qwen()
    cur.wait_stream(side)
    return f
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print(f"qwen {t(qwen):.1f} ms, encoder {t(enc):.1f} ms, sequential {t(seq):.1f} ms, two streams {t(par):.1f} ms")
