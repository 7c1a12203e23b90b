#!/usr/bin/env python3
"""One steady-state frame of the SAM2 stream (bench.py --mode sam2_stream) from a rocprofv3 --kernel-trace CSV: frames are delimited by the launches of an anchor
kernel that runs once per frame (default: the memory encoder's first conv, conv3x3s2_direct); the LAST full frame is listed launch by launch (start offset, duration,
gap to the previous launch) and summarised by kernel name, so dependent-chain latency (gaps) can be told from kernel time.
  python3 tools/frame_timeline.py <dir-or-csv> [--anchor NAME] [--group-ms G] [--list]      (G: anchor launches closer than this belong to one frame; 1.5 by default,
  0.3 for an anchor that runs once per frame now that a frame is shorter than 1.5 ms; --frame N lists the frame opened by the N-th anchor group instead of the last)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    src = sys.argv[1]
    anchor = arg("--anchor", "attn_split_combine")
    group_ns = float(arg("--group-ms", "1.5")) * 1e6
    path = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[2]]
    # frame boundary = first anchor launch after a stretch without one (the anchor may run several times per frame: group launches < 1.5 ms apart)
    starts = []
    for i in marks:
        if not starts or rows[i][0] - rows[starts[-1][1]][0] > group_ns:
            starts.append([i, i])
        else:
            starts[-1][1] = i
    if len(starts) < 3:
        print("fewer than three frames found for anchor", anchor)
        return
    per = [rows[starts[k + 1][0]][0] - rows[starts[k][0]][0] for k in range(len(starts) - 1)]
    per.sort()
    print(f"{len(starts)} anchor groups; period median {per[len(per) // 2] / 1e6:.3f} ms, min {per[0] / 1e6:.3f}, max {per[-1] / 1e6:.3f}")
    pick = int(arg("--frame", "-3"))     # which anchor group opens the listed frame (default: the last full one; e.g. 40 = a steady-state frame of the first timed pass)
    a, b = starts[pick][0], starts[pick + 1][0]
    fr = rows[a:b]
    t0 = fr[0][0]
    span = rows[b][0] - t0
    busy = sum(e - s for s, e, _ in fr)
    print(f"frame: {span / 1e6:.3f} ms, {len(fr)} launches, busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %)")
    agg = defaultdict(lambda: [0, 0.0])
    prev = None
    for s, e, n in fr:
        k = n.split("(")[0][:100]
        agg[k][0] += 1
        agg[k][1] += e - s
        if "--list" in sys.argv:
            print(f"  +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {((s - prev) / 1e3 if prev else 0):6.1f}  {k[:90]}")
        prev = e
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {t / 1e3:9.1f} us  x{c:3d}  avg {t / c / 1e3:7.1f}  {k}")


if __name__ == "__main__":
    main()
