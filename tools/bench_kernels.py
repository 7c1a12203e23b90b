"""Micro-benchmarks of the HIP kernels at the BASELINE.json config-2 shapes (run on the GPU box).

usage: python tools/bench_kernels.py [gemm] [attn] [rows]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).to(torch.bfloat16)


def bench_gemm(out):
    shapes = [
        ("vit.qkv", 8192, 3840, 1280), ("vit.proj", 8192, 1280, 1280), ("vit.gateup", 8192, 6912, 1280),
        ("vit.down", 8192, 1280, 3456), ("vit.patch", 8192, 1280, 1216), ("mrg.fc1", 2048, 5120, 5120),
        ("mrg.fc2", 2048, 3584, 5120), ("llm.qkv", 2112, 4608, 3584), ("llm.o", 2112, 3584, 3584),
        ("llm.gateup", 2112, 37888, 3584), ("llm.down", 2112, 3584, 18944), ("lm_head", 2112, 152064, 3584),
        ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192),
    ]
    for name, M, N, K in shapes:
        a, w = rnd(M, K), rnd(N, K, scale=0.02)
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for tile in (0, 1, 2, 10, 11, 12):
            ms = timeit(lambda: ops.gemm(a, w, out=c, tile=tile), iters=10 if M * N * K > 4e11 else 20)
            tf = 2.0 * M * N * K / ms / 1e9
            rec = {"op": "gemm", "name": name, "M": M, "N": N, "K": K, "tile": tile, "ms": round(ms, 4), "tflops": round(tf, 1)}
            print(json.dumps(rec), flush=True)
            out.append(rec)
        del a, w, c


def bench_attn(out):
    cases = [
        ("vit.win", [64] * 128, 16, 16, 80, False), ("vit.full", [1024] * 8, 16, 16, 80, False),
        ("llm.causal", [2112], 28, 4, 128, True), ("llm.causal4k", [4160], 28, 4, 128, True),
        ("hiera.w256", [256] * 16, 8, 8, 72, False), ("hiera.glob", [4096], 8, 8, 72, False),
        ("memattn", [4096], 1, 1, 256, False),
    ]
    for name, lens, Hq, Hkv, D, causal in cases:
        T = sum(lens)
        q, k, v = rnd(T, Hq, D), rnd(T, Hkv, D), rnd(T, Hkv, D)
        cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
        o = torch.empty_like(q)
        for impl in (0, 2):
            ms = timeit(lambda: ops.attn_varlen(q, k, v, cu, cu, max(lens), D ** -0.5, causal, out=o, impl=impl))
            fl = sum(4.0 * L * L * D * Hq for L in lens) * (0.5 if causal else 1.0)
            rec = {"op": "attn", "name": name, "impl": impl, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
            print(json.dumps(rec), flush=True)
            out.append(rec)


def bench_rows(out):
    for name, rows, dim in [("vit", 8192, 1280), ("llm", 2112, 3584)]:
        x, w = rnd(rows, dim), rnd(dim)
        ms = timeit(lambda: ops.rmsnorm(x, w, 1e-6))
        rec = {"op": "rmsnorm", "name": name, "ms": round(ms, 4), "GBps": round(rows * dim * 4 / ms / 1e6, 1)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    a, b = rnd(2112, 18944), rnd(2112, 18944)
    ms = timeit(lambda: ops.silu_mul(a, b))
    rec = {"op": "silu_mul", "ms": round(ms, 4), "GBps": round(a.numel() * 6 / ms / 1e6, 1)}
    print(json.dumps(rec), flush=True)
    out.append(rec)


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "attn", "rows"]
    res = []
    if "gemm" in what:
        bench_gemm(res)
    if "attn" in what:
        bench_attn(res)
    if "rows" in what:
        bench_rows(res)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_kernels.json"), "w") as f:
        json.dump(res, f, indent=1)
