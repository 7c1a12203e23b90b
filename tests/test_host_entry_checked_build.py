"""CPU: the host-only C entry points (Pillow coefficient tables, Qwen normalisation table: csrc/host_tables.cpp, plain C++ with no device code) run under the
CPU memory / undefined-behaviour checkers by the recipe in sanitize/ (SURVEY.md 5.2).  That directory is CPU-only and does not travel to the GPU box
(.gpurunignore), so this test skips there; it is not a gpu test.  The harness allocates exact-capacity buffers, so any write past the end of the caller's
tables aborts; too-small capacities and over-wide filters must be refused with an error code."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECIPE = os.path.join(ROOT, "sanitize")


@pytest.mark.skipif(not os.path.isdir(RECIPE) or shutil.which("gcc") is None, reason="CPU-only recipe not present (GPU box) or no gcc")
def test_host_entry_points_checked_build(tmp_path):
    r = subprocess.run(["make", "--no-print-directory", "-C", RECIPE, "check", f"OUT={tmp_path}"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])
