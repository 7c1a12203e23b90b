"""CPU: pins the joint-model oracle (oracle/unigr.py) against golden vectors from the reference's own UniGRModel.model_forward /
evaluate (tiny Qwen from transformers 5.15 + tiny SAM2 from the reference classes): five loss scalars, gradients of trainable
tensors, integer offsets, bool masks."""
import numpy as np
import pytest
import torch

from oracle import unigr as U
from tests.qwen_tiny import oracle_cfg
from tests.unigr_tiny import CASES, SEG, gold, make_batch, params, sam_cfg

GRAD_KEYS = ("text_hidden_fcs.0.2.weight", "lm_head.weight")


@pytest.fixture(scope="module")
def G():
    return gold()


@pytest.mark.parametrize("case", ["11", "10", "00"])
def test_model_forward_losses_and_grads(G, case):
    P, PS = params(G)
    for k in GRAD_KEYS:
        P[k].requires_grad_(True)
    sk = "sam_mask_decoder.output_hypernetworks_mlps.1.layers.2.weight"
    PS[sk].requires_grad_(True)
    b = make_batch(CASES[case], seed=int(case, 2) + 1)
    assert np.array_equal(b["input_ids"].numpy(), G[f"train_{case}_input_ids"])
    o = U.model_forward(P, PS, oracle_cfg(), sam_cfg(), b, (1.0, 0.5, 2.0), SEG)
    for k in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss"):
        assert abs(float(o[k]) - float(G[f"train_{case}_{k}"])) < 2e-5 * max(1.0, abs(float(G[f"train_{case}_{k}"]))), k
    o["loss"].backward()
    for k in GRAD_KEYS:
        if f"train_{case}_grad::{k}" not in G.files:   # no [SEG] anywhere: the reference produces no gradient for this tensor
            assert P[k].grad is None or float(P[k].grad.abs().max()) == 0.0
            continue
        ref = G[f"train_{case}_grad::{k}"]
        assert np.abs(P[k].grad.numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), k
    gk = f"train_{case}_grad::grounding_encoder.sam2_model.{sk}"
    if gk in G.files:
        assert np.abs(PS[sk].grad.numpy() - G[gk]).max() < 1e-5 * max(1e-3, np.abs(G[gk]).max())
    exp_off = np.concatenate([[0], np.cumsum(CASES[case])])
    assert np.array_equal(o["seg_token_offset"].numpy(), exp_off)  # integer plumbing bit-exact


def test_evaluate_bool_masks(G):
    P, PS = params(G)
    b = make_batch((True,), seed=9)
    assert np.array_equal(b["input_ids"].numpy(), G["eval_input_ids"])
    with torch.no_grad():
        _, masks, off, _ = U.evaluate(P, PS, oracle_cfg(), sam_cfg(), b, SEG, [(20, 28)])
    assert len(masks) == int(G["eval_n_masks"]) and off.tolist() == [0, 1]
    assert np.array_equal(masks[0].numpy(), G["eval_masks"])
