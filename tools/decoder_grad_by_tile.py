#!/usr/bin/env python3
"""Which GEMM tiling moves the mask decoder's token-path gradients?  Runs tests/test_fullsize_parity_gpu.py::_mask_decoder_case (firm ReLU point) with every tuned
product forced onto one tiling at a time, and untuned / tuned, printing the largest errors.  Diagnostic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd")); sys.path.insert(0, ROOT)
import torch
from rga3.hip import tuner
from tests.test_fullsize_parity_gpu import _mask_decoder_case
dev = torch.device("cuda:0")
def show(tag, r):
    e = r["errs"]
    print(f"{tag:10s} lang {e['language_embd']:.4f}  mask_tok {e['sam_mask_decoder.mask_tokens.weight']:.4f}  final_q {e['sam_mask_decoder.transformer.final_attn_token_to_image.q_proj.weight']:.4f}"
          f"  hyp0 {e['sam_mask_decoder.output_hypernetworks_mlps.2.layers.0.weight']:.4f}  up0 {e['sam_mask_decoder.output_upscaling.0.weight']:.4f}  low {r['low']:.4f}", flush=True)
for rep in range(3):
    tuner._cache.clear()
    show(f"tuned#{rep}", _mask_decoder_case(dev, firm_relu=True))
    print("   picks:", {k[:4]: v for k, v in tuner._cache.items() if v not in (-1,)}, flush=True)
for t in (-1, 20, 21, 22, 31, 32, 12, 13, 3, 4, 5, 14, 25):
    try:
        with tuner.force(t):
            show(f"tile {t}", _mask_decoder_case(dev, firm_relu=True))
    except Exception as ex:
        print(f"tile {t}: {type(ex).__name__} {str(ex)[:120]}", flush=True)
