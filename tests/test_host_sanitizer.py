"""CPU: the host-only C entry points (Pillow coefficient tables, Qwen normalisation table: csrc/preproc.hip) under AddressSanitizer + UBSan
(the GPU side cannot be sanitised on this pool).  The harness allocates exact-capacity buffers, so any write past the end of the
caller's tables aborts; too-small capacities and over-wide filters must be refused with an error code."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_host_entry_points_under_asan_ubsan(tmp_path):
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-fPIC"]
    objs = []
    for src in ("preproc.hip", "core.hip"):
        o = str(tmp_path / (src + ".o"))
        subprocess.run([HIPCC, *flags, "-c", os.path.join(ROOT, "rga3-release_amd", "csrc", src), "-o", o], check=True, timeout=300)
        objs.append(o)
    h = str(tmp_path / "harness.o")
    subprocess.run([HIPCC, "-x", "c", "-O1", "-g", "-fsanitize=address,undefined", "-c", os.path.join(ROOT, "tests", "asan", "host_entry_harness.c"), "-o", h],
                   check=True, timeout=300)
    exe = str(tmp_path / "harness")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", h, *objs, "-o", exe], check=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])
