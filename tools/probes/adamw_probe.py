#!/usr/bin/env python3
"""AdamW on lm_head-sized parameters (545 M, 30 B per element): time and bytes/s; DBG_LIB=old runs the previous build for an A/B; results compared bit for bit when both run."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import lib as _lib
if os.environ.get("DBG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", "librga3_hip_%s.so" % os.environ["DBG_LIB"])
from rga3.hip import ops
dev = "cuda"
torch.manual_seed(0)
for n in (152064 * 3584 // 4, 152064 * 3584):
    p = (torch.randn(n, device=dev) * 0.02).to(torch.bfloat16)
    master = p.float(); g = (torch.randn(n, device=dev) * 1e-3).to(torch.bfloat16)
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    ss = (g.float() ** 2).sum().reshape(1)
    for s in range(1, 3):
        ops.adamw_step_clip_(p, master, g, m, v, 1e-4, 0.9, 0.95, 1e-8, 0.0, s, ss, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in range(3, 8):
        ops.adamw_step_clip_(p, master, g, m, v, 1e-4, 0.9, 0.95, 1e-8, 0.0, s, ss, 1.0)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"n {n/1e6:7.1f} M: {ms*1e3:8.1f} us  {n*30/ms/1e9:6.2f} TB/s   checksum {float(master.double().sum()):.9e} {float(m.double().abs().sum()):.9e} {int(p.view(torch.int16).long().sum())}", flush=True)
    del p, master, g, m, v
