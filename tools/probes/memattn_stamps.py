#!/usr/bin/env python3
"""s_memtime stamps of workgroup 0 of the (measurement-build) staggered memory cross-attention kernel: per half-step start / after the phase / after the LDS store /
(next start = after the barrier).  The variant is not in the tree: git apply tools/probes/memattn_staggered_variant.patch && make -C rga3-release_amd/csrc, run this,
then git checkout rga3-release_amd/csrc/memattn.hip && make again."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
dev = "cuda"
buf = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
os.environ["RGA3_MA_TS"] = str(buf.data_ptr())
from rga3.hip import ops  # noqa: E402

torch.manual_seed(0)
q = torch.randn(4096, 256, device=dev).to(torch.bfloat16)
k = torch.randn(28736, 256, device=dev).to(torch.bfloat16)
m = torch.randn(28736, 64, device=dev).to(torch.bfloat16)
for _ in range(3):
    ops.memattn_cross(q, k, m, 256 ** -0.5, partials=True)
torch.cuda.synchronize()
buf.zero_()
ops.memattn_cross(q, k, m, 256 ** -0.5, partials=True)
torch.cuda.synchronize()
ts = buf.cpu().view(8, 1024)
t00 = int(ts[:, 0].min())
for w in (0, 4):
    row = ts[w]
    nst = int((row != 0).sum())
    print(f"wave {w}: {nst} stamps; start {int(row[0]) - t00}")
    for hs in range(10, 18):
        a, b, c_, d = (int(row[3 * hs + i]) for i in range(4))
        print(f"  hs {hs:2d}: phase {b - a:6d}  store {c_ - b:6d}  barrier {d - c_:6d}   total {d - a:6d}")
tot = int(ts[0][:171].max()) - t00
print("whole loop (wave 0):", tot, "ticks")
