import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import bench
dev = torch.device("cuda:0")
model, cfg, batch = bench.build_full(dev, 0, 16)
model.train()
for n, p in model.named_parameters():
    p.requires_grad_("sam_mask_decoder" in n or "text_hidden_fcs" in n)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model(**batch)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out["mask_loss"].backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for p in model.parameters():
        p.grad = None
print(json.dumps({"forward_ms": round((t1 - t0) * 1e3, 1), "mask_path_backward_ms": round((t2 - t1) * 1e3, 1)}))
from rga3.hip import tuner
for k, v in tuner.timings().items():
    if k[0] <= 1 and k[2] >= 4096:
        print("TUNE", k, tuner.table().get(k), {t: round(ms * 1e3, 1) for t, ms in v.items()})
