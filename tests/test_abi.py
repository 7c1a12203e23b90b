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
