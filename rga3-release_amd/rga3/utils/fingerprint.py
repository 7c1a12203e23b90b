"""Fingerprint of everything that decides WHICH kernels run and WHAT they do: the HIP sources and the package's Python.  Counter profiles (rocprofv3 --pmc passes
collected outside the timed run) carry the fingerprint of the tree they were measured on; bench.py refuses to quote a profile whose fingerprint is not the running
tree's (a traffic figure from another tree is not evidence for this one)."""
import hashlib
import os

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # .../rga3-release_amd


def tree_fingerprint() -> str:
    h = hashlib.sha256()
    files = []
    for sub, exts in (("csrc", (".hip", ".h", ".inc", ".cpp")), ("rga3", (".py",))):
        for d, _, fs in os.walk(os.path.join(_PKG, sub)):
            if "build" in d.split(os.sep) or "__pycache__" in d:
                continue
            files += [os.path.join(d, f) for f in fs if f.endswith(exts) or f == "Makefile"]
    for f in sorted(files):
        h.update(os.path.relpath(f, _PKG).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
