"""Condense a rocprofv3 --kernel-trace --stats run: python tools/kernel_stats_summary.py <dir> <needle> <out.json>
Sums the kernels whose name contains <needle> (the GEMM family: gemm_nt_kernel / gemm_nt_pp_kernel / gemm_nt_sk_kernel)."""
import csv, glob, json, sys
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rga3-release_amd"))
from rga3.utils.fingerprint import tree_fingerprint
d, needle, out = sys.argv[1:4]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
fam = [r for r in rows if needle in r["Name"]]
calls = sum(int(r["Calls"]) for r in fam)
ns = sum(float(r["TotalDurationNs"]) for r in fam)
res = {"needle": needle, "calls": calls, "total_ms": ns / 1e6, "avg_launch_ms": ns / 1e6 / max(calls, 1), "share_of_kernel_time": ns / tot,
       "all_kernels_total_ms": tot / 1e6, "tree": tree_fingerprint(),
       "top": [{"name": r["Name"][:110], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_us": float(r["AverageNs"]) / 1e3} for r in rows[:25]]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "top"}))
