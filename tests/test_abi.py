"""CPU: the C-ABI library loads and exports every symbol include/rga3_hip.h declares; the ctypes table matches the header."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    src = open(os.path.join(ROOT, "include", "rga3_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint(?:64_t)?\s+(rga3_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        decls[m.group(1)] = args
    return decls


def test_header_matches_ctypes_table():
    from rga3.hip import lib

    decls = header_decls()
    assert set(decls) == set(lib.SIGNATURES), set(decls) ^ set(lib.SIGNATURES)
    for name, args in decls.items():
        assert len(args) == len(lib.SIGNATURES[name]), name
        for a, ct in zip(args, lib.SIGNATURES[name]):
            if "*" in a:
                assert ct in (ctypes.c_void_p, ctypes.c_char_p), (name, a)
            elif a.startswith("int64_t"):
                assert ct is ctypes.c_int64, (name, a)
            elif a.startswith("float"):
                assert ct is ctypes.c_float, (name, a)
            elif a.startswith("size_t"):
                assert ct is ctypes.c_size_t, (name, a)
            else:
                assert ct is ctypes.c_int, (name, a)


def test_library_exports_all_symbols():
    from rga3.hip import lib

    assert os.path.exists(lib.LIB_PATH), "build the extension first (__graft_entry__.build())"
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in header_decls():
        assert hasattr(so, name), name
    assert lib.load().rga3_version() >= 1


def test_cpu_tensor_is_rejected_loudly():
    import pytest
    import torch

    from rga3.hip import lib, ops

    with pytest.raises(lib.Rga3Error):
        ops.rmsnorm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16), 1e-6)


def test_device_code_has_no_swapped_half_packed_f32():
    """The shipped code objects hold no v_pk_{mul,fma,add}_f32 whose op_sel swaps a source's halves: those forms (the SLP pass's pairing of independent
    f32 lanes, e.g. the RoPE rotation of memlayer_rows) returned wrong lanes 48-63 when the kernel shared a CU with other work -- the concurrent
    object-slot graphs' mismatch of round 6 (DESIGN 4 erratum note, tools/probes/coexec_probe3.py).  csrc/Makefile builds with -fno-slp-vectorize."""
    import glob
    import shutil
    import subprocess
    import tempfile

    from rga3.hip import lib

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    assert os.path.exists(objdump) and os.path.exists(lib.LIB_PATH)
    with tempfile.TemporaryDirectory() as d:
        so = shutil.copy(lib.LIB_PATH, os.path.join(d, "lib.so"))
        subprocess.run([objdump, "--offloading", so], cwd=d, check=True, capture_output=True, timeout=120)
        objs = sorted(glob.glob(os.path.join(d, "lib.so.*gfx950")))
        assert len(objs) >= 10, objs
        packed = swapped = 0
        for o in objs:
            text = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True, timeout=300).stdout
            for line in text.splitlines():
                if re.search(r"\bv_pk_\w+_f32\b", line):
                    packed += 1
                    m = re.search(r"op_sel:\[([01,]+)\]", line)
                    if m and "1" in m.group(1):
                        swapped += 1
        assert packed > 1000      # the explicit float2 / float4 code (attention rescale, epilogues) is still packed: the grep sees the code objects
        assert swapped == 0, swapped
