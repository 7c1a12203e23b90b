#!/usr/bin/env python3
"""The full-size mask-decoder gradient comparison (firm ReLU point) over several positional-matrix seeds: HIP path vs fp32 oracle, and -- oracle vs oracle -- what
bf16 storage of activations / gradients alone costs at the same point.  Diagnostic for DESIGN.md 2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd")); sys.path.insert(0, ROOT)
import torch
from tests.test_fullsize_parity_gpu import _mask_decoder_case
dev = torch.device("cuda:0")
keys = {"lang": "language_embd", "mask_tok": "sam_mask_decoder.mask_tokens.weight", "final_q": "sam_mask_decoder.transformer.final_attn_token_to_image.q_proj.weight",
        "hyp0": "sam_mask_decoder.output_hypernetworks_mlps.2.layers.0.weight", "mlp0": "sam_mask_decoder.transformer.layers.0.mlp.layers.0.weight", "up0": "sam_mask_decoder.output_upscaling.0.weight"}
for seed in range(1, 9):
    r = _mask_decoder_case(dev, firm_relu=True, pe_seed=seed, emulate=True)
    tot = sum(v * v for v in r["norms"].values()) ** 0.5
    sig = [n for n in r["errs"] if r["norms"][n] > 1e-5 * tot]
    print(f"seed {seed}: HIP " + " ".join(f"{k} {r['errs'][n]:.4f}" for k, n in keys.items()) + f" | max over significant {max(r['errs'][n] for n in sig):.4f}", flush=True)
    print(f"        EMU " + " ".join(f"{k} {r['emu'][n]:.4f}" for k, n in keys.items()) + f" | max over significant {max(r['emu'][n] for n in sig):.4f}", flush=True)
