"""Registers, spills, scratch and LDS of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).
usage: python tools/kernel_resources.py gemm_bf16 [name-filter] [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rga3-release_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-ffp-contract=fast",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src + ".hip"), "-o", "/tmp/_kr.o"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    recs, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            recs.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    dm = demangle([r["name"] for r in recs])
    print(f"{'VGPR':>5} {'AGPR':>5} {'vspill':>6} {'sspill':>6} {'scratch':>8} {'occ':>4} {'LDS':>7}  kernel")
    for r in recs:
        name = dm.get(r["name"], r["name"])
        if flt and flt not in name:
            continue
        print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>6} {r.get('SGPRs Spill', '?'):>6} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>8} {r.get('Occupancy [waves/SIMD]', '?'):>4} {r.get('LDS Size [bytes/block]', '?'):>7}  {name[:150]}")


if __name__ == "__main__":
    main()
