#!/usr/bin/env python3
"""Screen + rates for tile 26 of rga3_gemm_bf16 (persistent 256 x 256 + stream-K tail whose RAGGED last tile row -- <= 64 rows, M = 2112 = 8 x 256 + 64 -- runs the
quarter-work loop, the tail's runs sized by cost).  A sync-structure edit makes a new template (cdna_hip_programming.md 5): screened over many runs at several sizes.
  * tile 27 (no K split) BIT-identical to tile 21; tile 26 against tile 21 equal to rounding (K-split tiles sum in another order);
  * tile 26 identical run to run;
  * rates of tiles 22 / 26 / 32 (/ 31) on the model's M = 2112 and M = 4160 products, interleaved.
python3 tools/probes/ragged_probe.py [screen|time|both] [repeats]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SCREEN = [(2112, 4608, 3584, "none", False), (2112, 3584, 3584, "none", True), (2112, 37888, 3584, "swiglu", False), (2112, 3584, 18944, "none", True),
          (2112, 18944, 3584, "none", False), (2112, 3584, 37888, "none", False), (2112, 8192, 3584, "gelu", False), (2112, 152064, 3584, "none", False),
          (320, 256, 64, "none", False), (320, 512, 128, "none", True), (257, 256, 192, "relu", False), (300, 700, 448, "none", True), (2050, 1000, 3584, "gelu", True),
          (4160, 4608, 3584, "none", False), (4160, 3584, 18944, "none", True), (4160, 37888, 3584, "swiglu", False), (577, 512, 6400, "none", False),
          (2113, 3584, 3584, "none", True), (1088, 16384, 1024, "none", False), (66000 // 256 * 256 + 40, 1280, 1280, "none", True)]
TIME = [(2112, 37888, 3584, "swiglu", False), (2112, 3584, 18944, "none", True), (2112, 4608, 3584, "none", False), (2112, 3584, 3584, "none", True),
        (2112, 18944, 3584, "none", False), (2112, 3584, 37888, "none", False), (2112, 152064, 3584, "none", False),
        (4160, 37888, 3584, "swiglu", False), (4160, 3584, 18944, "none", True), (4160, 4608, 3584, "none", False)]


def operands(M, N, K, act, res):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    nout = N // 2 if act == "swiglu" else N
    r = torch.randn(M, nout, device="cuda", generator=g).to(torch.bfloat16) if res else None
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    return a, w, bias, r


bad = 0
if what in ("screen", "both"):
    for M, N, K, act, res in SCREEN:
        a, w, bias, r = operands(M, N, K, act, res)
        ref = ops.gemm(a, w, bias, residual=r, act=act, tile=21)
        first = None
        nrep = nbad = 0
        info = info27 = ""
        for i in range(reps):
            z = ops.gemm(a, w, bias, residual=r, act=act, tile=26)
            z27 = ops.gemm(a, w, bias, residual=r, act=act, tile=27)
            if not torch.equal(z27, ref):     # no K split: one summation order per element, the order of tile 21
                nbad += 1
                info27 = f" [tile 27 != tile 21: {int((z27 != ref).sum())} elements]"
            if first is None:
                first = z
                zf, rf = z.float(), ref.float()
                e = float((zf - rf).norm() / rf.norm())
                fin = bool(torch.isfinite(zf).all())
                m_last = (M - 1) // 256 * 256
                neq_rag = int((z[m_last:] != ref[m_last:]).sum())
                neq_all = int((z != ref).sum())
                info = f"rel-L2 vs tile 21 {e:.2e}, elements differing {neq_all} ({neq_rag} of them in the last tile row's {M - m_last} rows)"
                if e > 2e-3 or not fin:
                    nbad += 1
            elif not torch.equal(z, first):
                nrep += 1
        bad += nbad + nrep
        print(f"M={M:<6d} N={N:<7d} K={K:<6d} {act:7s} res={int(res)}: {info}{info27}; off {nbad}, not reproducible in {nrep}/{reps - 1} runs", flush=True)
    torch.cuda.synchronize()
    print("stream-K give-ups:", ops.gemm_stream_k_timeouts(a.device))
    print("SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})", flush=True)

if what in ("time", "both"):
    tiles = (22, 26, 27, 32, 31, 21)
    for M, N, K, act, res in TIME:
        a, w, bias, r = operands(M, N, K, act, res)
        nout = N // 2 if act == "swiglu" else N
        out = torch.empty(M, nout, device="cuda", dtype=torch.bfloat16)
        best = {t: float("inf") for t in tiles}
        for rnd in range(3):
            for t in tiles:
                f = lambda: ops.gemm(a, w, bias, residual=r, act=act, out=out, tile=t)
                for _ in range(2):
                    f()
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(10):
                    f()
                en.record()
                en.synchronize()
                best[t] = min(best[t], st.elapsed_time(en) / 10 * 1e3)
        print(f"M={M:<6d} N={N:<7d} K={K:<6d} {act:7s} res={int(res)}  us (TF/s) by tile  " +
              "  ".join(f"{t}: {best[t]:7.1f} ({2.0 * M * N * K / best[t] / 1e6:5.0f})" for t in tiles), flush=True)
