"""attn_causal32_kernel against the general kernel (impl = 4) on the decoder's causal rows: interleaved rounds in one process, median / min.
  python3 tools/causal32_probe.py"""
import os, sys, json, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

dev = "cuda"
res = []
for (S, Hq, Hkv) in ((2112, 28, 4), (4160, 28, 4), (8192, 28, 4)):
    D = 128
    torch.manual_seed(0)
    qkv = torch.randn(S, Hq + 2 * Hkv, D, device=dev).to(torch.bfloat16)
    q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:]
    cu = torch.tensor([0, S], dtype=torch.int32, device=dev)
    out = torch.empty(S, Hq, D, dtype=torch.bfloat16, device=dev)
    t = {0: [], 4: []}
    for rnd in range(12):
        for impl in (0, 4):
            for _ in range(3):
                ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, True, out=out, impl=impl)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(20):
                ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, True, out=out, impl=impl)
            en.record(); en.synchronize()
            t[impl].append(st.elapsed_time(en) / 20 * 1e3)
    fl = 4.0 * S * S * D * Hq / 2
    line = {"S": S, "Hq": Hq, "Hkv": Hkv, "causal32_us_median": round(statistics.median(t[0]), 1), "causal32_us_min": round(min(t[0]), 1),
            "general_us_median": round(statistics.median(t[4]), 1), "general_us_min": round(min(t[4]), 1),
            "causal32_tf": round(fl / statistics.median(t[0]) / 1e6, 1), "general_tf": round(fl / statistics.median(t[4]) / 1e6, 1)}
    print(json.dumps(line), flush=True)
    res.append(line)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r04_causal32_probe.json"), "w"), indent=1)
