#!/usr/bin/env python3
"""bench.py against another build of the library (A/B on one box): python3 tools/probes/bench_with_lib.py <suffix|-> <bench args...>
suffix 'old' -> rga3-release_amd/librga3_hip_old.so; '-' -> the product library."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
suffix = sys.argv[1]
from rga3.hip import lib as _lib
if suffix != "-":
    import ctypes
    import torch  # noqa: F401  (its HIP runtime first: the library must bind to that one)
    _lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", f"librga3_hip_{suffix}.so")
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in [n for n in _lib.SIGNATURES if not hasattr(so, n)]:     # an older build lacks the entry points (and tilings) added since
        del _lib.SIGNATURES[name]
    if not hasattr(so, "rga3_gemm_ragged_plan"):
        from rga3.hip import tuner as _tuner
        _tuner.CANDIDATES = tuple(t for t in _tuner.CANDIDATES if t not in (26, 27))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
