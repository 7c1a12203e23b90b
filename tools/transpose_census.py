#!/usr/bin/env python3
"""Which tensors does one training step transpose (ops.transpose), and how long does each take?  python3 tools/transpose_census.py"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from rga3.hip import lib, ops
    lib.load()
    model, cfg, inputs = bench.build_full(dev, 0, 16)
    trainables, reducer, opt = bench.make_trainable(model, True)

    def step():
        reducer.begin_step()
        reducer.begin_micro_step()
        model(**inputs)["loss"].backward()
        reducer.finish()
        opt.step(reducer.grad_view, reducer.flat_grads())

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    real = ops.transpose
    rec = []

    def tr(x):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = real(x)
        e.record()
        rec.append((tuple(x.shape), s, e))
        return r

    ops.transpose = tr
    try:
        step()
    finally:
        ops.transpose = real
    torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    for shp, s, e in rec:
        agg[shp][0] += 1
        agg[shp][1] += s.elapsed_time(e)
    for shp, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{ms:7.3f} ms {c:4d} x {shp}")


if __name__ == "__main__":
    main()
