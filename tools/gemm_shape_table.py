#!/usr/bin/env python3
"""Per-shape table of the GEMM-family launches of one training step (BASELINE configs[2] per GPU): HIP events around every ops.gemm / gemm_tn call, grouped by
(kind, M, N, K, epilogue); prints launches per step, ms per step and TFLOP/s by shape, sorted by time -- which products to work on.
  python3 tools/gemm_shape_table.py [out.json]"""
import json
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
        out = model(**inputs)
        out["loss"].backward()
        reducer.finish()
        opt.step(reducer.grad_view, reducer.flat_grads())

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    rec = []
    real_gemm, real_tn = ops.gemm, ops.gemm_tn

    def gemm(a, w, bias=None, residual=None, act="none", **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = real_gemm(a, w, bias, residual=residual, act=act, **k)
        e.record()
        rec.append((("nt", a.shape[0], w.shape[0], a.shape[1], act + ("+res" if residual is not None else "") + ("+f32" if r.dtype == torch.float32 else "")), s, e))
        return r

    def gemm_tn(a, b, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = real_tn(a, b, **k)
        e.record()
        rec.append((("tn", a.shape[1], b.shape[1], a.shape[0], ""), s, e))
        return r

    real_ln = ops.gemm_ln

    def gemm_ln(a, stats, wf, colc, biasf, act="none", **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = real_ln(a, stats, wf, colc, biasf, act=act, **k)
        e.record()
        rec.append((("nt", a.shape[0], wf.shape[0], a.shape[1], "ln+" + act), s, e))
        return r

    real_cat, real_many, real_pre = ops.gemm_cat, ops.gemm_tn_many, ops.gemm_swiglu_pre

    def gemm_cat(a, w, bias=None, a2=None, w2=None, wn=None, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = real_cat(a, w, bias, a2=a2, w2=w2, wn=wn, **k)
        e.record()
        # one launch: [a | a2] [w | w2]^T (K + K2) and a wn^T (N2 more columns); the row reports the flop-equivalent N x K of both sides
        K2, N2 = (a2.shape[1] if a2 is not None else 0), (wn.shape[0] if wn is not None else 0)
        rec.append((("nt", a.shape[0], w.shape[0] + N2, a.shape[1] + K2, "cat" + ("+K%d" % K2 if K2 else "") + ("+N%d" % N2 if N2 else "")), s, e,
                    2.0 * a.shape[0] * (w.shape[0] * (a.shape[1] + K2) + N2 * a.shape[1])))
        return r

    def gemm_tn_many(pairs, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = real_many(pairs, **k)
        e.record()
        rec.append((("tn", sum(x.shape[1] for x, _ in pairs), pairs[0][1].shape[1], pairs[0][0].shape[0], "x%d grouped" % len(pairs)), s, e,
                    sum(2.0 * x.shape[1] * y.shape[1] * x.shape[0] for x, y in pairs)))
        return r

    def gemm_swiglu_pre(a, w, bias=None, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = real_pre(a, w, bias, **k)
        e.record()
        rec.append((("nt", a.shape[0], w.shape[0], a.shape[1], "swiglu+pre"), s, e))
        return r

    real_lnsum = ops.gemm_lnsum

    def gemm_lnsum(a, w, bias=None, residual=None, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = real_lnsum(a, w, bias, residual=residual, **k)
        e.record()
        rec.append((("nt", a.shape[0], w.shape[0], a.shape[1], "none" + ("+res" if residual is not None else "") + "+lnsum"), s, e))
        return r

    ops.gemm, ops.gemm_tn, ops.gemm_ln, ops.gemm_lnsum = gemm, gemm_tn, gemm_ln, gemm_lnsum
    ops.gemm_cat, ops.gemm_tn_many, ops.gemm_swiglu_pre = gemm_cat, gemm_tn_many, gemm_swiglu_pre
    nst = 3
    try:
        for _ in range(nst):
            step()
    finally:
        ops.gemm, ops.gemm_tn, ops.gemm_ln, ops.gemm_lnsum = real_gemm, real_tn, real_ln, real_lnsum
        ops.gemm_cat, ops.gemm_tn_many, ops.gemm_swiglu_pre = real_cat, real_many, real_pre
    torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for it in rec:
        key, s, e = it[:3]
        agg[key][0] += 1
        agg[key][1] += s.elapsed_time(e)
        agg[key][2] += it[3] if len(it) > 3 else 2.0 * key[1] * key[2] * key[3]
    rows = []
    for (kind, M, N, K, epi), (c, ms, fl) in agg.items():
        rows.append({"kind": kind, "M": M, "N": N, "K": K, "epi": epi, "launches_per_step": c / nst, "ms_per_step": ms / nst, "tflops": fl / (ms * 1e-3) / 1e12})
    rows.sort(key=lambda r: -r["ms_per_step"])
    tot = sum(r["ms_per_step"] for r in rows)
    print(f"GEMM family: {tot:.2f} ms per step over {sum(r['launches_per_step'] for r in rows):.0f} launches")
    for r in rows[:60]:
        print(f"  {r['ms_per_step']:7.3f} ms {r['launches_per_step']:6.1f} x  {r['kind']} M={r['M']:<8d} N={r['N']:<7d} K={r['K']:<8d} {r['epi']:<12s} {r['tflops']:7.0f} TF/s")
    if len(sys.argv) > 1:
        json.dump({"total_ms_per_step": tot, "shapes": rows}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
