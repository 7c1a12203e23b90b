"""UniGRModel.evaluate() end to end at 7B + SAM2-L (random weights): one 16-frame clip, one [SEG].  python tools/evaluate_probe.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import bench
dev = torch.device("cuda:0")
model, cfg, b = bench.build_full(dev, 0, 16)
model.eval()
kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], pixel_values=None, pixel_values_videos=b["pixel_values_videos"], image_grid_thw=None,
          video_grid_thw=b["video_grid_thw"], second_per_grid_ts=b.get("second_per_grid_ts"), images_sam=b["images_sam"], resize_list=None,
          original_size_list=[(480, 640)])
def run():
    out, masks = model.evaluate(**kw)
    return masks
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): m = run()
torch.cuda.synchronize()
print(f"evaluate(): {(time.perf_counter()-t0)/5*1e3:.1f} ms per 16-frame clip; masks {[tuple(x.shape) for x in m]}")
