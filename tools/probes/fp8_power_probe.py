"""Is the e4m3 product power-bound like the bf16 ones?  Board power and shader clock (bench.py's in-process sysfs sampler) while the configs[4] decoder shapes run in a
loop as e4m3 (gemm_fp8) and as bf16 (gemm), with the achieved rate of each.  python3 tools/probes/fp8_power_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import bench  # noqa: E402
from rga3.hip import ops  # noqa: E402


def rate(f, flops, n=30):
    for _ in range(5):
        f()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        f()
    en.record()
    en.synchronize()
    return flops / (st.elapsed_time(en) / n * 1e-3) / 1e12


for M, N, K in ((4160, 37888, 3584), (4160, 3584, 18944), (4160, 4608, 3584), (16384, 6912, 1280)):
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    aq, sa = ops.quant_fp8_rows(a)
    wq, sw = ops.quant_fp8_rows(w)
    o8 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f8 = lambda: ops.gemm_fp8(aq, sa, wq, sw, out=o8)
    fb = lambda: ops.gemm(a, w, out=ob)
    for name, f in (("e4m3", f8), ("bf16", fb)):
        tf = rate(f, 2.0 * M * N * K)

        def loop():
            for _ in range(20):
                f()
        board = bench._board_sample(loop, seconds=2.5)
        pw = board.get("power_w") if board else None
        ck = board.get("sclk_mhz") if board else None
        print(f"M={M:<6d} N={N:<6d} K={K:<6d} {name}: {tf:7.0f} TFLOP/s   power {pw}   sclk {ck}", flush=True)
    torch.cuda.synchronize()
    time.sleep(0.5)
