#!/usr/bin/env python3
"""Does a re-read operand stay in the Infinity Cache?  A streaming read kernel (rga3_sumsq_det: 16 B per lane, 2048 workgroups) is run repeatedly over
buffers of growing size; the achieved bytes/s over a footprint that fits the 256 MiB Infinity Cache vs one that does not shows which reads reach HBM.
(Context: rocprofv3's FETCH_SIZE counts fabric-side requests, Infinity-Cache hits included — MI355X_MICROARCH.md — so the 4-5x "traffic" of the GEMM
family over its algorithmic bytes cannot be attributed by counters; gfx950 exposes no MALL / HBM counter: gpurun_out/counters_list.txt.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import torch  # noqa: E402

from rga3.hip import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    part = torch.zeros(2048, dtype=torch.float32, device=dev)
    acc = torch.zeros(1, dtype=torch.float32, device=dev)
    rows = []
    for mb in (8, 15, 32, 64, 128, 200, 400, 1024, 4096):
        n = mb * (1 << 20) // 2
        x = torch.ones(n, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.sumsq_det_(x, part, acc, False)
        reps = max(5, min(200, 20000 // mb))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ops.sumsq_det_(x, part, acc, False)
        e.record()
        e.synchronize()
        ms = s.elapsed_time(e) / reps
        rows.append({"buffer_MiB": mb, "us_per_pass": round(ms * 1e3, 2), "TB_per_s": round(n * 2 / (ms * 1e-3) / 1e12, 2)})
        print(rows[-1], flush=True)
        del x
    json.dump(rows, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None


if __name__ == "__main__":
    main()
