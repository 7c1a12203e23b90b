"""A/B of one module switch on a bench.py mode, two fresh processes per side, interleaved:   python3 tools/probes/ab_bench.py rga3.model.sam2=_LN_SUMS train_full [bench args]
Each side runs `bench.py --mode <mode> --no-cpu-baseline --no-board <args>` with the named module attribute set True / False before main(); prints ms_per_step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec, mode, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
mod, attr = spec.split("=")
code = ("import sys, importlib; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench; m = importlib.import_module(%r); setattr(m, %r, {val}); "
        "sys.argv = ['bench.py', '--mode', %r, '--no-cpu-baseline', '--no-board'] + %r; bench.main()") % (ROOT, os.path.join(ROOT, "rga3-release_amd"), mod, attr, mode, extra)
res = {"True": [], "False": []}
for rnd in range(2):
    for val in ("True", "False"):
        env = dict(os.environ, RGA3_BENCH_TIMED_ONLY="1")
        r = subprocess.run([sys.executable, "-c", code.format(val=val)], capture_output=True, text=True, env=env, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("run failed:", r.stderr[-1500:])
            continue
        res[val].append(json.loads(line[-1])["ms_per_step"])
        print(f"{attr}={val} round {rnd}: {res[val][-1]} ms", flush=True)
print("A/B", spec, mode, {k: v for k, v in res.items()})
