"""ORACLE — test infrastructure only.  fp32 restatement of the joint model's seam: reference
model/qwen_2_5_vl_sam2.py:149-393 (model_forward / evaluate) composed from oracle/qwen25vl.py and oracle/sam2.py.
Pinned by tests/test_oracle_unigr.py against golden vectors produced by the reference's own UniGRModel
(tests/golden/make_unigr_fixtures.py)."""
import torch
import torch.nn.functional as F

from . import qwen25vl as Q
from . import sam2 as S


def dice_loss(inputs, targets, num_masks, scale=1000, eps=1e-6):
    """reference :17-40"""
    inputs = inputs.sigmoid().flatten(1, 2)
    targets = targets.flatten(1, 2)
    num = 2 * (inputs / scale * targets).sum(-1)
    den = (inputs / scale).sum(-1) + (targets / scale).sum(-1)
    return (1 - (num + eps) / (den + eps)).sum() / (num_masks + 1e-8)


def sigmoid_ce_loss(inputs, targets, num_masks):
    """reference :43-60"""
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.flatten(1, 2).mean(1).sum() / (num_masks + 1e-8)


def seg_embeddings(P, hidden, seg_token_mask):
    """reference :212-218 text_hidden_fcs on all positions then boolean gather."""
    h = F.linear(F.relu(F.linear(hidden, P["text_hidden_fcs.0.0.weight"], P["text_hidden_fcs.0.0.bias"])),
                 P["text_hidden_fcs.0.2.weight"], P["text_hidden_fcs.0.2.bias"])
    return h[seg_token_mask]


def shifted_seg_mask(ids, seg_idx):
    m = ids == seg_idx
    return torch.cat([m[:, 1:], torch.zeros_like(m)[:, 0].unsqueeze(1)], dim=1)


def model_forward(P, PS, qcfg, scfg, batch, weights, seg_token_idx, out_dim=256, inference=False, internals=None):
    """reference :149-321.  P: Qwen+head params, PS: SAM2 params (names without the 'grounding_encoder.sam2_model.' prefix).
    weights = (ce, dice, bce).  inference=True is the branch validate() drives (reference :236-257, train_joint.py:586-648): batch size 1
    (the reference squeezes dim 0 of images_sam), SAM2 video inference with the language prompt on every frame, bilinear to the label size,
    sigmoid > 0.5; a sample without [SEG] is prompted with the zero embedding."""
    images_sam = batch["images_sam"].float()
    B, T = images_sam.shape[:2]
    r = Q.forward(P, qcfg, batch["input_ids"], batch.get("attention_mask"), position_ids=batch.get("position_ids"), labels=batch["labels"],
                  pixel_values_videos=batch["pixel_values_videos"].float(), video_grid_thw=batch["video_grid_thw"],
                  second_per_grid_ts=batch.get("second_per_grid_ts"))
    ce_loss = r["loss"] * weights[0]
    mask = shifted_seg_mask(batch["labels"], seg_token_idx)
    pred = seg_embeddings(P, r["hidden"], mask)
    counts = mask.int().sum(-1)
    off = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(-1)])[batch["offset"]]
    embs = []
    for i in range(len(off) - 1):
        a, b = int(off[i]), int(off[i + 1])
        embs += [torch.zeros(1, out_dim) if a == b else pred[a:b]] * T
    lang = torch.cat(embs, dim=0).unsqueeze(1)
    if inference:
        assert B == 1, "reference :243 squeezes the batch dimension"
        le = lang.reshape(B, T, 1, out_dim)
        pred_masks, logits = [], []
        for i in range(B):
            masks, _ = S.language_embd_inference(PS, images_sam[0], [le[i][t] for t in range(T)], scfg)
            m = F.interpolate(masks, size=tuple(batch["label_list"][i].shape), mode="bilinear", align_corners=False)[:, 0]
            pred_masks.append(m.sigmoid() > 0.5)
            logits.append(m)
        return {"pred_masks": pred_masks, "gt_masks": batch["masks_list"], "mask_logits": logits, "seg_token_offset": off}
    feats = S.prepare_backbone_features(S.image_encoder_forward(PS, images_sam.flatten(0, 1), scfg))
    if internals is not None and lang.requires_grad:
        lang.retain_grad()
        internals["lang"] = lang
    _, high, _ = S.inject_language_embd_train(PS, feats, lang, scfg, internals)
    if internals is not None:
        high.retain_grad()
        internals["high"] = high
    high = high.reshape(B, T, scfg.image_size, scfg.image_size)
    bce = dice = 0
    n_tot = 0
    has_seg = counts.bool()
    for i in range(B):
        pm = F.interpolate(high[i].unsqueeze(1), size=batch["label_list"][i].shape, mode="bilinear", align_corners=False)[:, 0]
        gt = batch["masks_list"][i]
        if not has_seg[i]:
            pm = pm[0:0]
        assert gt.shape[0] == pm.shape[0]
        bce = bce + sigmoid_ce_loss(pm, gt, gt.shape[0]) * gt.shape[0]
        dice = dice + dice_loss(pm, gt, gt.shape[0]) * gt.shape[0]
        n_tot += gt.shape[0]
    bce = weights[2] * bce / (n_tot + 1e-8)
    dice = weights[1] * dice / (n_tot + 1e-8)
    return {"loss": ce_loss + bce + dice, "ce_loss": ce_loss, "mask_bce_loss": bce, "mask_dice_loss": dice, "mask_loss": bce + dice,
            "seg_token_offset": off, "pred_embeddings": pred}


def evaluate(P, PS, qcfg, scfg, batch, seg_token_idx, original_size_list):
    """reference :325-393"""
    r = Q.forward(P, qcfg, batch["input_ids"], batch.get("attention_mask"), position_ids=batch.get("position_ids"),
                  pixel_values_videos=batch["pixel_values_videos"].float(), video_grid_thw=batch["video_grid_thw"],
                  second_per_grid_ts=batch.get("second_per_grid_ts"))
    mask = shifted_seg_mask(batch["input_ids"], seg_token_idx)
    pred = seg_embeddings(P, r["hidden"], mask)
    counts = mask.int().sum(-1)
    off = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(-1)])
    out, logits = [], []
    imgs = batch["images_sam"][0].float()
    for i in range(len(off) - 1):
        e = pred[int(off[i]):int(off[i + 1])]
        masks, _ = S.language_embd_inference(PS, imgs, [e] * imgs.shape[0], scfg)
        h, w = original_size_list[i]
        m = F.interpolate(masks, size=(h, w), mode="bilinear", align_corners=False)[:, 0]
        out.append(m.sigmoid() > 0.5)
        logits.append(m)
    return r, out, off, logits
