"""Ablation of attn_causal32_kernel's loop (library built with `make AB=1`): RGA3_C32_DBG mask 1 = no tile DMA in the loop, 2 = no softmax, 4 = no P V, 8 = no K Q^T,
16 = no wait / barrier.  Timing only (outputs of the ablated variants are garbage)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
    from rga3.hip import ops
    S, Hq, Hkv, D = 2112, 28, 4, 128
    torch.manual_seed(0)
    qkv = torch.randn(S, Hq + 2 * Hkv, D, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:]
    cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
    out = torch.empty(S, Hq, D, dtype=torch.bfloat16, device="cuda")
    ts = []
    for rnd in range(8):
        for _ in range(5):
            ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, True, out=out)
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(20):
            ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, True, out=out)
        en.record(); en.synchronize()
        ts.append(st.elapsed_time(en) / 20 * 1e3)
    ts.sort()
    print(json.dumps({"mask": int(os.environ.get("RGA3_C32_DBG", "0")), "us_median": round(ts[len(ts) // 2], 1), "us_min": round(ts[0], 1)}))
else:
    for m in (0, 1, 2, 4, 8, 6, 10, 12, 14, 15, 16, 17, 31):
        env = dict(os.environ, RGA3_C32_DBG=str(m))
        r = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
