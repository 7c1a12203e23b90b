#!/usr/bin/env python3
"""Board power and shader clock while one GEMM tiling runs in a loop (rocm-smi sampled from a side thread): evidence for DESIGN lesson 7 (MFMA-busy x clock is constant
across the tilings: the large products are power-bound).  python3 tools/probes/power_clock_probe.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}
    d = {}
    m = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([\d.]+)", out)
    if m:
        d["power_w"] = float(m.group(1))
    m = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
    if m:
        d["sclk_mhz"] = int(m.group(1))
    m = re.search(r"Temperature \(Sensor junction\) \(C\):\s*([\d.]+)", out)
    if m:
        d["junction_c"] = float(m.group(1))
    if not d:
        d["raw"] = out[-600:]
    return d


CASES = [("gate|up (2112,37888,3584) swiglu", 2112, 37888, 3584, "swiglu", (22, 27, 21)), ("8192^3", 8192, 8192, 8192, "none", (20, 21, 28)),
         ("Hiera fc1 (65536,2304,576) gelu", 65536, 2304, 576, "gelu", (20,)), ("idle", 0, 0, 0, "none", (0,))]
for name, M, N, K, act, tiles in CASES:
    for tile in tiles:
        samples = []
        stop = [False]

        def poll():
            time.sleep(1.0)
            while not stop[0]:
                samples.append(smi())
                time.sleep(0.5)

        th = threading.Thread(target=poll)
        th.start()
        t0 = time.time()
        n = 0
        if M:
            a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
            out = torch.empty(M, N // 2 if act == "swiglu" else N, device="cuda", dtype=torch.bfloat16)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            while time.time() - t0 < 5.0:
                for _ in range(50):
                    ops.gemm(a, w, act=act, out=out, tile=tile)
                n += 50
                torch.cuda.synchronize()
            en.record()
            en.synchronize()
            us = st.elapsed_time(en) / n * 1e3
            tf = 2.0 * M * N * K / us / 1e6
        else:
            time.sleep(4.0)
            us = tf = 0.0
        stop[0] = True
        th.join()
        pw = [s["power_w"] for s in samples if "power_w" in s]
        ck = [s["sclk_mhz"] for s in samples if "sclk_mhz" in s]
        tj = [s["junction_c"] for s in samples if "junction_c" in s]
        print(f"{name:34s} tile {tile:2d}: {us:8.1f} us {tf:6.0f} TFLOP/s | power W {pw} | sclk MHz {ck} | junction C {tj}" + ("" if pw or ck else f" | {samples[-1] if samples else None}"), flush=True)
