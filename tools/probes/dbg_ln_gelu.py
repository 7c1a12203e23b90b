import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import torch.nn.functional as F
from rga3.hip import lib as _lib
if os.environ.get("DBG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", "librga3_hip_%s.so" % os.environ["DBG_LIB"])
from rga3.hip import ops
print("lib", _lib.LIB_PATH)
dev = "cuda"
M, N, K = 4096, 2304, 576
g = torch.Generator().manual_seed(M + N)
x = (torch.randn(M, K, generator=g) * 0.7 + torch.randn(M, 1, generator=g) * 1.5).to(torch.bfloat16).to(dev)
w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
b = (torch.randn(N, generator=g) * 0.2).to(torch.bfloat16).to(dev)
gamma = (1 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16).to(dev)
beta = (0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).to(dev)
st = ops.layernorm_stats(x, 1e-6)
wf, colc, bf = ops.fold_layernorm(w, b, gamma, beta)
xf = x.float().cpu()
ref0 = F.layer_norm(xf, (K,), gamma.float().cpu(), beta.float().cpu(), 1e-6) @ w.float().cpu().t() + b.float().cpu()
for act in ("none", "gelu"):
    ref = F.gelu(ref0) if act == "gelu" else ref0
    for tile in (-1, 5, 20, 21):
        for rep in range(2):
            out = ops.gemm_ln(x, st, wf, colc, bf, act=act, tile=tile).float().cpu()
            d = (out - ref)
            rel = float(d.norm() / ref.norm())
            bad = (d.abs() > 0.05 + 0.02 * ref.abs())
            nb = int(bad.sum())
            msg = f"{act} tile {tile} rep {rep}: rel_l2 {rel:.5f} bad {nb}"
            if nb:
                rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
                msg += f" rows {rows[:8].tolist()}..{int(rows[-1])} (n {len(rows)}) cols {cols[:8].tolist()}..{int(cols[-1])} (n {len(cols)})"
                r, c = bad.nonzero()[0].tolist()
                msg += f" e.g. [{r},{c}] out {float(out[r, c]):.4f} ref {float(ref[r, c]):.4f} pre {float(ref0[r, c]):.4f}"
                # histogram of bad by (col % 192) / 16 and row % 128 / 16
                bc = torch.bincount((bad.nonzero()[:, 1] % 192) // 16, minlength=12).tolist()
                br = torch.bincount((bad.nonzero()[:, 0] % 128) // 16, minlength=8).tolist()
                msg += f" by ntile {bc} by mtile {br}"
            print(msg, flush=True)
