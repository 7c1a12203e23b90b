#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient errors of the SAM2-L mask decoder at full size (tests/test_fullsize_parity_gpu.py::_mask_decoder_case) and the errors of the
gradients of the head's intermediate activations, HIP path vs fp32 oracle.   python3 tools/decoder_fullsize_grad.py [feature_scale]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests.test_fullsize_parity_gpu import _mask_decoder_case, rel  # noqa: E402

dbg = {}
r = _mask_decoder_case(torch.device("cuda:0"), float(sys.argv[1]) if len(sys.argv) > 1 else 1.0, dbg, firm_relu=len(sys.argv) > 2 and sys.argv[2] == "firm")
print("best", r["best"], "low", r["low"], "loss", r["loss"])
tot = sum(v * v for v in r["norms"].values()) ** 0.5
for n, e in sorted(r["errs"].items(), key=lambda kv: -kv[1])[:60]:
    print(f"{e:9.4f}  |g|/|G| {r['norms'][n] / tot:9.2e}  {n}")
d, ri = dbg["product"], dbg["oracle"]
B = ri["masks"].shape[0]
h = ri["upscaled"].shape[2] // 4
def cmp(name, got, gg, ref, rg):
    print(f"{name:8s} value rel {rel(got.detach(), ref.detach()):.4f}  grad rel {rel(gg, rg):.4f}  |grad| {float(rg.float().norm()):.3e}", flush=True)
cmp("masks", d["masks"], d["masks"].grad, ri["masks"], ri["masks"].grad)
cmp("hyper", d["hyper"], d["hyper"].grad, ri["hyper"], ri["hyper"].grad)
t2m = lambda t: t.float().view(B, 4 * h, 4 * h, -1).permute(0, 3, 1, 2)
cmp("up", t2m(d["up"].detach()), t2m(d["up"].grad), ri["upscaled"], ri["upscaled"].grad)
cmp("hs", d["hs"], d["hs"].grad, ri["hs"], ri["hs"].grad)
s2t = lambda t: t.float().view(B, -1, t.shape[-1])
cmp("src", s2t(d["src"].detach()), s2t(d["src"].grad), ri["src"].flatten(2).transpose(1, 2), ri["src"].grad.flatten(2).transpose(1, 2))
cmp("tokens", s2t(d["tokens"].detach()), s2t(d["tokens"].grad), ri["tokens"], ri["tokens"].grad)
