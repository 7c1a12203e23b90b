"""Where one full RGA3 training step spends its time (sync-bracketed phases; adds sync overhead, use for proportions)."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import bench
from rga3.model.qwen_train import add_lora
dev = torch.device("cuda:0")
model, cfg, batch = bench.build_full(dev, 0, 16)
add_lora(model, r=128, alpha=256, dropout=0.05, exclude=("sam_model", "grounding_encoder", "visual", "text_hidden_fcs"))
model.train()
for n, p in model.named_parameters():
    p.requires_grad_(("lora_" in n) or n in ("lm_head.weight", "model.embed_tokens.weight") or ("sam_mask_decoder" in n or "text_hidden_fcs" in n))
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[key] = T.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
ge = model.grounding_encoder
wrap(ge, "get_sam2_embeddings_train", "sam2_encoder_fwd")
wrap(ge, "inject_language_embd_train", "mask_decoder_fwd")
for it in range(4):
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model(**batch)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out["mask_loss"].backward(retain_graph=True)
    torch.cuda.synchronize(); t15 = time.perf_counter()
    out["ce_loss"].backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for p in model.parameters():
        p.grad = None
print(json.dumps({"forward_ms": round((t1 - t0) * 1e3, 1), "backward_mask_loss_ms": round((t15 - t1) * 1e3, 1), "backward_ce_loss_ms": round((t2 - t15) * 1e3, 1), **{k: round(v * 1e3, 1) for k, v in T.items()}}))
