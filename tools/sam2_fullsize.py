"""SAM2-L at full size on one GPU (random-init weights): training-path forward over F frames and the memory-attention video
stream (BASELINE.json configs[3]: prompt on frame 0 only, then propagate).  Prints timings as JSON lines."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.model.sam2 import SAM2, VideoSession  # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = SAM2().to(torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.normal_(0, 0.02)
    n_par = sum(p.numel() for p in m.parameters())
    img = torch.randn(F, 3, 1024, 1024, device=dev).to(torch.bfloat16)
    emb = torch.randn(F, 1, 256, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st = m.get_sam2_embeddings_train(img)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            low, high = m.inject_language_embd_train(st, emb)
            torch.cuda.synchronize(); t2 = time.perf_counter()
        assert torch.isfinite(high).all()
        print(json.dumps({"what": "sam2_train_path_forward", "frames": F, "params": n_par, "encoder_ms_per_frame": round((t1 - t0) * 1e3 / F, 3),
                          "heads_ms_per_frame": round((t2 - t1) * 1e3 / F, 3), "encoder_tflops": round(1.82 * F / (t1 - t0), 1),
                          "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)
        vid = torch.randn(T, 3, 1024, 1024, device=dev).to(torch.bfloat16)
        for it in range(2):
            sess = VideoSession(m.sam2_model, vid)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sess.add_language_embd(0, emb[:1])
            res = sess.propagate()
            torch.cuda.synchronize(); t1 = time.perf_counter()
        print(json.dumps({"what": "sam2_memory_stream", "frames": T, "ms_per_frame": round((t1 - t0) * 1e3 / T, 3), "frames_per_s": round(T / (t1 - t0), 2),
                          "counts": sess.counts}), flush=True)


if __name__ == "__main__":
    main()
