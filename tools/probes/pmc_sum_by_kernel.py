"""Per-kernel sums of rocprofv3 --pmc counter_collection CSVs under a directory: python3 tools/probes/pmc_sum_by_kernel.py <dir> [name-filter]
Prints, per kernel name, launches and the per-launch average of every counter found (FETCH_SIZE / WRITE_SIZE as the guide prescribes: KB, FETCH x2 on gfx950)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "").split("(")[0]
            if flt and flt not in name:
                continue
            c, v = r.get("Counter_Name"), float(r.get("Counter_Value") or 0)
            acc[name][c] += v
            cnt[name][c] += 1
for name in sorted(acc):
    parts = []
    for c in sorted(acc[name]):
        n = cnt[name][c]
        v = acc[name][c] / max(n, 1)
        if c == "FETCH_SIZE":
            parts.append(f"fetch {v * 1024 * 2 / 1e6:.1f} MB")
        elif c == "WRITE_SIZE":
            parts.append(f"write {v * 1024 / 1e6:.1f} MB")
        else:
            parts.append(f"{c} {v:.3g}")
    n0 = max(cnt[name].values())
    print(f"{name[:70]:70s} x{n0:4d}  " + "  ".join(parts))
