"""Fingerprint of everything that decides WHICH kernels run and WHAT they do: the HIP sources, the package's Python, the BUILT library that actually runs (a stale
build of edited sources must not pass), and the scripts that define the measured shapes and derive the traffic figures (bench.py, tools/pmc_*.py).  Counter profiles (rocprofv3 --pmc passes
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
    root = os.path.dirname(_PKG)
    tools = os.path.join(root, "tools")
    files += [os.path.join(root, "bench.py")] + ([os.path.join(tools, f) for f in os.listdir(tools) if f.startswith("pmc_") and f.endswith(".py")] if os.path.isdir(tools) else [])
    files.append(os.path.join(_PKG, "librga3_hip.so"))      # the binary the process maps
    for f in sorted(files):
        if not os.path.exists(f):
            h.update(b"<missing>" + os.path.relpath(f, root).encode())
            continue
        h.update(os.path.relpath(f, _PKG).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
