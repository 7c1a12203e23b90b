#!/usr/bin/env python3
"""Hiera-L stage-3 products (K = 576, 65 536 tokens = 16 frames; 32 768 = 8) by the activation-stationary kernel against the tuned LDS-tiled kernels.
python3 tools/gemm_xstat_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rga3.hip import ops  # noqa: E402
from memlayer_probe import t_us  # noqa: E402


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for M in (65536, 32768):
        x = torch.randn(M, 576, device=dev).to(torch.bfloat16)
        st = ops.layernorm_stats(x, 1e-6)
        for name, N, act, res, ln in (("qkv (LN fold)", 1728, "none", False, True), ("proj + residual", 576, "none", True, False), ("fc1 (LN fold) + gelu", 2304, "gelu", False, True)):
            w = (torch.randn(N, 576, device=dev) * 0.05).to(torch.bfloat16)
            b = torch.randn(N, device=dev).to(torch.bfloat16)
            r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
            colc = w.float().sum(1).contiguous()
            fl = 2.0 * M * N * 576
            if ln:
                t_old = t_us(lambda: ops.gemm_ln(x, st, w, colc, b, act=act), n=5, reps=5)
                t_new = t_us(lambda: ops.gemm_xstat(x, w, b, None, act, st, colc), n=5, reps=5)
                err = ((ops.gemm_xstat(x, w, b, None, act, st, colc).float() - ops.gemm_ln(x, st, w, colc, b, act=act).float()).norm() / ops.gemm_ln(x, st, w, colc, b, act=act).float().norm()).item()
            else:
                t_old = t_us(lambda: ops.gemm(x, w, b, residual=r, act=act), n=5, reps=5)
                t_new = t_us(lambda: ops.gemm_xstat(x, w, b, r, act), n=5, reps=5)
                err = ((ops.gemm_xstat(x, w, b, r, act).float() - ops.gemm(x, w, b, residual=r, act=act).float()).norm() / ops.gemm(x, w, b, residual=r, act=act).float().norm()).item()
            print(f"M={M:6d} {name:22s} N={N:5d}: tiled {t_old:7.1f} us ({fl / t_old / 1e6:6.0f} TF/s)   activation-stationary {t_new:7.1f} us ({fl / t_new / 1e6:6.0f} TF/s)   rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
