#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: the step is the span between the last two bursts of the optimizer kernel
(--anchor, default adamw8_kernel); it is cut into --bin-ms bins and each bin prints busy %, launches and the kernel that holds most of its time, so a
host-bound stretch (many launches, low busy %) can be told from a kernel-bound one.
--from-ms / --to-ms additionally list the kernels of that stretch of the step by total time.
  python3 tools/step_timeline.py <dir-or-csv> [--anchor adamw8_kernel] [--bin-ms 2.5] [--from-ms 108 --to-ms 134] [--exclude gemm_]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    src = sys.argv[1]
    anchor = arg("--anchor", "adamw8_kernel")
    bin_ns = float(arg("--bin-ms", "2.5")) * 1e6
    path = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # bursts of the anchor kernel: consecutive anchor launches less than 20 ms apart belong to one optimizer step
    ends = []
    last = None
    for s, e, n in rows:
        if anchor in n:
            if last is None or s - last > 20e6:
                ends.append(e)
            else:
                ends[-1] = e
            last = e
    if len(ends) < 2:
        print("fewer than two optimizer bursts in the trace")
        return
    t0, t1 = ends[-2], ends[-1]
    step = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    busy = sum(e - s for s, e, _ in step)
    print(f"step {(t1 - t0) / 1e6:.3f} ms, {len(step)} launches, busy {busy / 1e6:.3f} ms ({100 * busy / (t1 - t0):.1f} %)")
    nb = int((t1 - t0) / bin_ns) + 1
    bt = [0.0] * nb
    bc = [0] * nb
    bk = [defaultdict(float) for _ in range(nb)]
    for s, e, n in step:
        b = int((s - t0) / bin_ns)
        bc[b] += 1
        k = n.split("(")[0].replace("void ", "").replace("rga3::", "")[:60]
        x = s
        while x < e:   # spread a long kernel over the bins it covers
            bi = int((x - t0) / bin_ns)
            lim = min(e, t0 + (bi + 1) * bin_ns)
            bt[bi] += lim - x
            bk[bi][k] += lim - x
            x = lim
    for b in range(nb):
        top = max(bk[b].items(), key=lambda kv: kv[1])[0] if bk[b] else "-"
        print(f"  {b * bin_ns / 1e6:7.1f} ms  busy {100 * bt[b] / bin_ns:5.1f} %  launches {bc[b]:4d}  {top}")
    if "--from-ms" in sys.argv:
        stretch(step, t0, float(arg("--from-ms", "0")), float(arg("--to-ms", "1e9")))


def stretch(step, t0, a_ms, b_ms):
    excl = arg("--exclude", None)   # e.g. --exclude gemm_ : everything but the GEMM family
    sel = [r for r in step if r[0] >= t0 + a_ms * 1e6 and r[0] < t0 + b_ms * 1e6 and not (excl and excl in r[2])]
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        k = n.split("(")[0].replace("void ", "").replace("rga3::", "")[:90]
        agg[k][0] += e - s
        agg[k][1] += 1
    busy = sum(v[0] for v in agg.values())
    print(f"stretch {a_ms}..{b_ms} ms: {len(sel)} launches, busy {busy / 1e6:.3f} ms")
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f"  {t / 1e6:8.3f} ms {c:5d} x {t / c / 1e3:8.2f} us  {k}")


if __name__ == "__main__":
    main()
