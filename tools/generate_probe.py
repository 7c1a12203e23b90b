"""Decode-step rate of generate() at 7B dims (random weights): prefill S = 2112 (16 frames), then N new tokens.  python tools/generate_probe.py [N]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev)
inputs = bench.make_inputs(cfg, dev)
with torch.no_grad():
    for n in (8, N, N):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.generate(**inputs, max_new_tokens=1, do_sample=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out = model.generate(**inputs, max_new_tokens=n + 1, do_sample=False)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"prefill+1: {(t1-t0)*1e3:.1f} ms; {n} more tokens: {((t2-t1)-(t1-t0))*1e3/n:.2f} ms/token; out {tuple(out.shape)}", flush=True)
