#!/usr/bin/env python3
"""Hiera-L stage-3 products (K = 576 family, 16 frames = 65 536 rows): time per tiling, and a fixed launch sequence for rocprofv3 --pmc passes.

  python3 tools/probes/k576_probe.py time [out.json]     every (shape, tiling): median of 7 timed groups of 4 launches (HIP events on the launch stream)
  python3 tools/probes/k576_probe.py pmc                  3 launches of every (shape, tiling) of PMC_CASES, in that order (dispatch order identifies them)
  python3 tools/probes/k576_probe.py sum <dir> <out.json> reads *counter_collection.csv under <dir>, groups the gemm_nt launches by 3 in dispatch order

Shapes (reference model/sam2.py:986-1117 MultiScaleBlock at dim 576, :2305 MLP): fc1 = LayerNorm-folded 576 -> 2304 + GELU, fc2 = 2304 -> 576 + residual,
qkv = LayerNorm-folded 576 -> 1728, proj = 576 -> 576 + residual."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = 65536
SHAPES = {"fc1": (2304, 576, "ln+gelu"), "qkv": (1728, 576, "ln"), "fc2": (576, 2304, "res"), "proj": (576, 576, "res")}
TILES = {"fc1": (5, 3, 6, 20, -1), "qkv": (5, 3, 6, 20, -1), "fc2": (5, 12, 3, 20, 21, 23), "proj": (5, 12, 3, 20, 23)}
PMC_CASES = [("fc1", 5), ("fc1", 20), ("qkv", 5), ("qkv", 20), ("fc2", 5), ("fc2", 20), ("proj", 5)]


def setup():
    import torch
    sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
    from rga3.hip import ops
    torch.manual_seed(0)
    dev, bf = "cuda", torch.bfloat16
    rn = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(bf)
    x576 = rn(M, 576)
    x2304 = rn(M, 2304)
    st = ops.layernorm_stats(x576, 1e-6)
    data = {}
    for name, (N, K, kind) in SHAPES.items():
        w = rn(N, K, scale=0.04)
        b = rn(N)
        if kind.startswith("ln"):
            g, be = rn(K) + 1, rn(K, scale=0.1)
            wf, colc, bfold = ops.fold_layernorm(w, b, g, be)
            data[name] = (wf, colc, bfold)
        else:
            data[name] = (w, b)
    outs = {N: torch.empty((M, N), dtype=bf, device=dev) for N in (2304, 1728, 576)}
    res = rn(M, 576)

    def call(name, tile):
        N, K, kind = SHAPES[name]
        if kind == "ln+gelu":
            return ops.gemm_ln(x576, st, *data[name], act="gelu", out=outs[N], tile=tile)
        if kind == "ln":
            return ops.gemm_ln(x576, st, *data[name], act="none", out=outs[N], tile=tile)
        a = x2304 if K == 2304 else x576
        return ops.gemm(a, data[name][0], bias=data[name][1], residual=res, out=outs[N], tile=tile)
    return torch, call


def time_mode(out):
    torch, call = setup()
    rows = []
    for name, (N, K, kind) in SHAPES.items():
        fl = 2.0 * M * N * K
        for tile in TILES[name]:
            try:
                for _ in range(3):
                    call(name, tile)
                torch.cuda.synchronize()
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(4):
                        call(name, tile)
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / 4)
                ts.sort()
                us = ts[len(ts) // 2] * 1e3
                rows.append(dict(shape=name, N=N, K=K, kind=kind, tile=tile, us=round(us, 1), tflops=round(fl / us / 1e6, 1)))
                print(f"{name:5s} N={N:5d} K={K:5d} {kind:8s} tile {tile:3d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
            except Exception as e:   # noqa: BLE001
                print(f"{name} tile {tile}: {str(e)[:120]}", flush=True)
    if out:
        json.dump(rows, open(out, "w"), indent=1)


def pmc_mode():
    torch, call = setup()
    torch.cuda.synchronize()
    for name, tile in PMC_CASES:
        for _ in range(3):
            call(name, tile)
        torch.cuda.synchronize()


def sum_mode(d, out):
    per = defaultdict(dict)    # dispatch id -> counter -> value ; names
    names = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "gemm_nt" not in n:
                continue
            did = int(row["Dispatch_Id"])
            per[did][row["Counter_Name"]] = per[did].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            names[did] = n
    ids = sorted(per)
    # every pass replays the same sequence: dispatch ids are equal across passes (same program), so counters of different passes merge per id
    res = []
    for ci, (name, tile) in enumerate(PMC_CASES):
        grp = ids[3 * ci:3 * ci + 3]
        if len(grp) < 3:
            break
        N, K, kind = SHAPES[name]
        c = defaultdict(float)
        for g in grp[1:]:     # skip the first (cold) launch
            for k, v in per[g].items():
                c[k] += v / 2
        alg = (M * K + N * K + M * N + (M * N if kind == "res" else 0)) * 2
        r = dict(shape=name, tile=tile, kernel=names[grp[0]][:70], algorithmic_bytes=alg, counters={k: round(v, 1) for k, v in sorted(c.items())})
        if "FETCH_SIZE" in c:
            r["fetch_bytes_x2_corrected"] = c["FETCH_SIZE"] * 1024 * 2     # MI355X_MICROARCH.md: gfx950 FETCH_SIZE = half the bytes of wide coalesced reads
        if "WRITE_SIZE" in c:
            r["write_bytes"] = c["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            r["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), 4)
        if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            r["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        res.append(r)
        print(json.dumps(r))
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "time":
        time_mode(sys.argv[2] if len(sys.argv) > 2 else None)
    elif mode == "pmc":
        pmc_mode()
    else:
        sum_mode(sys.argv[2], sys.argv[3])
