#!/usr/bin/env python3
"""Any tools/ script against another build of the library (A/B on one box): python3 tools/probes/run_with_lib.py <suffix|-> <script> [args...]
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
    _lib.LIB_PATH = os.path.join(ROOT, "rga3-release_amd", f"librga3_hip_{suffix}.so")
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
