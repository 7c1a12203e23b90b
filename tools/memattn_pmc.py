#!/usr/bin/env python3
"""Only the memory cross-attention kernel at the stream's steady-state shape (4096 queries x 28 736 keys), a few launches: the target of a rocprofv3 --pmc pass.
python3 tools/memattn_pmc.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
q = torch.randn(4096, 256, device=dev).to(torch.bfloat16)
k = torch.randn(28736, 256, device=dev).to(torch.bfloat16)
m = torch.randn(28736, 64, device=dev).to(torch.bfloat16)
for _ in range(5):
    ops.memattn_cross(q, k, m, 256 ** -0.5, partials=True)
torch.cuda.synchronize()
print("ok")
