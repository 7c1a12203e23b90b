#!/usr/bin/env python3
"""Per-kernel time of the steady-state frames of the SAM2 stream (rocprofv3 --kernel-trace CSV): takes the last `--last-ms` of the trace and prints
busy %, launches and the kernels by total time.   python3 tools/frame_breakdown.py <dir> [--last-ms 40]"""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else 40.0
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
agg = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = n.split("(")[0][:80]
    agg[k][0] += e - s
    agg[k][1] += 1
print(f"window {span / 1e6:.3f} ms, {len(rows)} launches, busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %)")
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"  {t / 1e6:8.3f} ms {c:6d} x {t / c / 1e3:8.2f} us  {k}")
