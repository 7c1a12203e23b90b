#!/usr/bin/env python3
"""Where is the GPU idle inside a step?  Reads a rocprofv3 --kernel-trace CSV (columns Kernel_Name, Start_Timestamp, End_Timestamp), takes the last
`--steps` repetitions of the busiest periodic region (delimited by the kernel named with --anchor, default the LM-head-sized GEMM is not known here, so
the largest inter-kernel gaps delimit steps), and prints busy time, span, and the idle gaps grouped by the kernel that precedes them.
  python3 tools/trace_gaps.py <dir-or-csv> [--last-ms 300]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else 300.0
    path = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = defaultdict(lambda: [0, 0])
    big = []
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g > 0:
            key = n0.split("(")[0][:70]
            gaps[key][0] += g
            gaps[key][1] += 1
            if g > 50_000:
                big.append((g, n0[:60], n1[:60]))
    print(f"window {span / 1e6:.3f} ms, {len(rows)} launches, busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.3f} ms")
    print("idle time by preceding kernel (ms total, count, us avg):")
    for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {g / 1e6:8.3f} {c:6d} {g / c / 1e3:8.2f}  {k}")
    print("gaps > 50 us:", len(big))
    for g, a, b in sorted(big, reverse=True)[:15]:
        print(f"  {g / 1e3:9.1f} us  after {a}  before {b}")


if __name__ == "__main__":
    main()
