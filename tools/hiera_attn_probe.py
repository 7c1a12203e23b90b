#!/usr/bin/env python3
"""Hiera-L stage-3 window attention exactly as the model calls it (16 frames: 65 536 tokens, windows of 256, 8 heads x 72, q / k / v = slices of the packed
qkv rows), timed per launch for the kernel variants the launcher can pick, plus the global-attention block (4096 tokens per frame).  python3 tools/hiera_attn_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402


def t_us(f, n=20):
    for _ in range(3):
        f()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        f()
    en.record()
    en.synchronize()
    return st.elapsed_time(en) / n * 1e3


def main():
    dev = "cuda"
    N, heads, hd = 65536, 8, 72
    do = heads * hd
    qkv = torch.randn(N, 3 * do, device=dev).to(torch.bfloat16)
    kv = qkv.view(N, 3 * heads, hd)
    for seg, label in ((256, "windows of 256"), (4096, "global, 4096 per frame"), (64, "windows of 64")):
        cu = torch.arange(0, N + 1, seg, dtype=torch.int32, device=dev)
        fl = 4.0 * N * seg * do
        for impl in (0, 2):
            out = torch.empty(N, heads, hd, dtype=torch.bfloat16, device=dev)
            f = lambda: ops.attn_varlen(kv[:, :heads], kv[:, heads:2 * heads], kv[:, 2 * heads:], cu, cu, seg, hd ** -0.5, causal=False, out=out, impl=impl)
            us = t_us(f)
            print(f"{label:24s} packed qkv  impl={impl}: {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s", flush=True)
        if seg > 256:    # five of six output tiles (D = 72 <= 80) against all six (impl bit 16): same bits, fewer MFMAs
            ref = torch.empty(N, heads, hd, dtype=torch.bfloat16, device=dev)
            us = t_us(lambda: ops.attn_varlen(kv[:, :heads], kv[:, heads:2 * heads], kv[:, 2 * heads:], cu, cu, seg, hd ** -0.5, causal=False, out=ref, impl=16))
            print(f"{label:24s} packed qkv  six output tiles (impl=16): {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s   bit-equal to impl=0: {bool(torch.equal(ref, out))}", flush=True)
        if seg <= 256:   # whole-segment-in-LDS window kernel (needs the key range)
            out = torch.empty(N, heads, hd, dtype=torch.bfloat16, device=dev)
            us = t_us(lambda: ops.attn_varlen(kv[:, :heads], kv[:, heads:2 * heads], kv[:, 2 * heads:], cu, cu, seg, hd ** -0.5, causal=False, out=out, max_k=seg))
            print(f"{label:24s} packed qkv  window kernel: {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s", flush=True)
            us = t_us(lambda: ops.attn_varlen(kv[:, :heads], kv[:, heads:2 * heads], kv[:, 2 * heads:], cu, cu, seg, hd ** -0.5, causal=False, out=out, max_k=seg, impl=8))
            print(f"{label:24s} packed qkv  window kernel, 16 rows per wave (impl=8): {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s", flush=True)
        q, k, v = (kv[:, i * heads:(i + 1) * heads].contiguous() for i in range(3))
        out = torch.empty(N, heads, hd, dtype=torch.bfloat16, device=dev)
        us = t_us(lambda: ops.attn_varlen(q, k, v, cu, cu, seg, hd ** -0.5, causal=False, out=out))
        print(f"{label:24s} separate q/k/v impl=0: {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s", flush=True)
        # head dim padded to 96 in memory (what the kernel computes on anyway)
        qp = torch.zeros(N, heads, 96, dtype=torch.bfloat16, device=dev)
        kp, vp = torch.zeros_like(qp), torch.zeros_like(qp)
        qp[..., :hd], kp[..., :hd], vp[..., :hd] = q, k, v
        outp = torch.empty_like(qp)
        us = t_us(lambda: ops.attn_varlen(qp, kp, vp, cu, cu, seg, hd ** -0.5, causal=False, out=outp))
        print(f"{label:24s} D padded to 96     : {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
