"""CPU: pins the joint-model oracle (oracle/unigr.py) against golden vectors from the reference's own UniGRModel.model_forward /
evaluate (tiny Qwen from transformers 5.15 + tiny SAM2 from the reference classes): five loss scalars, gradients of trainable
tensors, integer offsets, bool masks."""
import numpy as np
import pytest
import torch

from oracle import unigr as U
from tests.qwen_tiny import oracle_cfg
from tests.unigr_tiny import CASES, LABEL_HW, SEG, gold, make_batch, params, sam_cfg

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
        _, masks, off, _ = U.evaluate(P, PS, oracle_cfg(), sam_cfg(), b, SEG, [LABEL_HW])
    assert len(masks) == int(G["eval_n_masks"]) and off.tolist() == [0, 1]
    assert np.array_equal(masks[0].numpy(), G["eval_masks"])


@pytest.mark.parametrize("tag,flags,seed", [("1", (True,), 11), ("0", (False,), 12)])
def test_model_forward_inference_branch(G, tag, flags, seed):
    """model_forward(inference=True) — the branch validate() drives (reference qwen_2_5_vl_sam2.py:236-257, train_joint.py:586-648): bool masks of the
    reference itself, with a [SEG] sample and with the zero-embedding prompt of a sample without one.  Clips are not part of the read-out fit."""
    P, PS = params(G)
    b = make_batch(flags, seed=seed)
    assert np.array_equal(b["input_ids"].numpy(), G[f"infer_{tag}_input_ids"])
    with torch.no_grad():
        o = U.model_forward(P, PS, oracle_cfg(), sam_cfg(), b, (1.0, 0.5, 2.0), SEG, inference=True)
    assert len(o["pred_masks"]) == 1 and o["pred_masks"][0].dtype == torch.bool
    ref = G[f"infer_{tag}_pred_masks"]
    got = o["pred_masks"][0].numpy()
    lg = o["mask_logits"][0].numpy()
    assert got.shape == ref.shape
    # fp32 vs fp32: identical except where the logit itself is at rounding distance from 0
    assert np.array_equal(got[np.abs(lg) > 1e-3], ref[np.abs(lg) > 1e-3]) and (got != ref).mean() < 1e-4
    if flags[0]:   # the mask is a blob with a real margin, not speckle: this is what makes the thresholded comparison meaningful
        assert (np.abs(lg) > 0.05 * np.abs(lg).max()).mean() > 0.97 and 0.02 < got.mean() < 0.6


def test_two_optimizer_steps_h1(G):
    """loss dict before each of two optimizer steps, pre-clip gradient norm, gradients and parameter deltas of four tensors, as the
    reference model + torch.optim.AdamW produced them (clip 1.0, lr 4e-5, betas (0.9, 0.95), wd 0; reference train_joint.py:300-324, 534-535): the oracle
    with plain torch AdamW must retrace them."""
    P, PS = params(G)
    names = [k for k in P if any(x in k for x in ("lm_head", "embed_tokens", "text_hidden_fcs"))]
    snames = [k for k in PS if k.startswith("sam_mask_decoder.")]
    train = [P[k].requires_grad_(True) for k in names] + [PS[k].requires_grad_(True) for k in snames]
    P0 = {k: P[k].detach().clone() for k in names}
    PS0 = {k: PS[k].detach().clone() for k in snames}
    opt = torch.optim.AdamW(train, lr=4e-5, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0)
    b = make_batch(CASES["11"], seed=4)
    pre = "grounding_encoder.sam2_model."
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        o = U.model_forward(P, PS, oracle_cfg(), sam_cfg(), b, (1.0, 0.5, 2.0), SEG)
        for k in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss"):
            r = float(G[f"h1_step{step}_{k}"])
            assert abs(float(o[k].detach()) - r) < 5e-5 * max(1.0, abs(r)), (step, k)
        o["loss"].backward()
        norm = float(torch.nn.utils.clip_grad_norm_([p for p in train if p.grad is not None], 1.0))
        assert abs(norm - float(G[f"h1_step{step}_grad_norm"])) < 2e-4 * norm, step
        opt.step()
        for key in [k for k in G.files if k.startswith(f"h1_delta{step + 1}::")]:
            n = key.split("::")[1]
            cur, old = (PS[n[len(pre):]], PS0[n[len(pre):]]) if n.startswith(pre) else (P[n], P0[n])
            d, ref = (cur.detach() - old).numpy(), G[key]
            # Adam's first steps are +-lr almost everywhere (|g| >> eps): elements whose gradient is at rounding level may take either sign
            close = np.abs(d - ref) <= 1e-6 + 0.02 * np.abs(ref)
            assert close.mean() > 0.995, (key, close.mean())
