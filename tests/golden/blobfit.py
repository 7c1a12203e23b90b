"""Fixture construction helper: gives the tiny random-weight SAM2 mask head REAL margins.

Random weights on random inputs produce "speckle" masks whose logits hover around zero, so a sigmoid > 0.5 test on them pins nothing
(VERDICT r1, weak item 1).  Here the last layer of the three multimask hyper-network MLPs (sam_mask_decoder.output_hypernetworks_mlps.{1,2,3}.layers.2,
3 x (32 x 256 + 32) numbers) is FITTED, by ridge regression in closed form, so that on the fixture clips (tests/blob_inputs.py: an ellipse drifting over a
smooth background) the mask logits are about +8 on the object and -8 off it (what a trained SAM2 emits), and the last bias of the IoU head is spread so
the argmax over the three candidates is decisive.  Everything upstream keeps its name-derived deterministic weights.  The fitted tensors are DATA:
they are stored in the .npz fixtures ("fit::<name>") and poured into the reference's classes (fixture generation), the oracle and the product alike
(tests/sam2_tiny.py, tests/unigr_tiny.py overlay them).  The fit runs on the oracle's restatement of the decoder (equal to the reference to 1e-5,
tests/test_oracle_sam2.py); the fixture OUTPUTS are then produced by the reference's own classes carrying these weights."""
import torch
import torch.nn.functional as F

from oracle import sam2 as S

HYPER = "sam_mask_decoder.output_hypernetworks_mlps.{}.layers.2.{}"
IOU_BIAS = "sam_mask_decoder.iou_prediction_head.layers.2.bias"


def hyper_hidden(P, tok, i):
    """Input of output_hypernetworks_mlps.i.layers.2 for mask-token rows tok [B, 256]."""
    pre = f"sam_mask_decoder.output_hypernetworks_mlps.{i}"
    return torch.relu(S.lin(torch.relu(S.lin(tok, P, pre + ".layers.0")), P, pre + ".layers.1"))


def candidate_targets(obj_masks_lowres, amp=8.0):
    """[B, h, w] float {0,1} at the decoder's low resolution -> [B, 3, h, w] targets: the object, a dilated and an eroded version."""
    m = obj_masks_lowres[:, None]
    dil, ero = F.max_pool2d(m, 3, 1, 1), -F.max_pool2d(-m, 3, 1, 1)
    return torch.cat([(x > 0.5).float() * 2 * amp - amp for x in (m, dil, ero)], 1)


def decoder_inputs(P, cfg, imgs):
    """(pix [B, C, s, s] incl. no_mem_embed, [feat_s0, feat_s1]) of the training path (reference sam2.py:343-375)."""
    with torch.no_grad():
        vf, _, sizes = S.prepare_backbone_features(S.image_encoder_forward(P, imgs, cfg))
    B = vf[-1].shape[1]
    high = [x.permute(1, 2, 0).view(x.shape[1], x.shape[2], *s) for x, s in zip(vf[:-1], sizes[:-1])]
    pix = (vf[-1] + P["no_mem_embed"]).permute(1, 2, 0).view(B, cfg.d_model, *sizes[-1])
    return pix, high


def fit(P, cfg, imgs, language_embd, obj_masks, lam=1e-3, iou_bias=(0.0, 1.5, 0.0, -1.5), chunk=4):
    """imgs [B, 3, S, S], language_embd [B, 1, 256], obj_masks bool [B, S, S].  Returns {name: tensor} of the fitted parameters.
    Ridge in the primal: logit[b, p] = sum_{c, j} W[c, j] z1[b, j] U[b, c, p], so A^T A = sum_b (U_b U_b^T) (x) (z1_b z1_b^T)."""
    sums = [None] * 3
    C = J = None
    for b0 in range(0, imgs.shape[0], chunk):
        sl = slice(b0, b0 + chunk)
        pix, high = decoder_inputs(P, cfg, imgs[sl])
        internals = {}
        with torch.no_grad():
            S.forward_sam_heads(P, pix, high, language_embd[sl], cfg, True, internals)
        toks, up = internals["mask_toks"], internals["upscaled"].double()      # [b, 4, 256], [b, 32, h, w]
        nb, C, h, w = up.shape
        low = F.adaptive_avg_pool2d(obj_masks[sl].float()[:, None], (h, w))[:, 0]
        tg = candidate_targets((low > 0.5).float()).double().reshape(nb, 3, -1)
        U = up.reshape(nb, C, -1)
        G = U @ U.transpose(1, 2)                                              # [b, C, C]
        for k in range(3):
            z = hyper_hidden(P, toks[:, k + 1], k + 1).double()
            z1 = torch.cat([z, torch.ones(nb, 1, dtype=torch.float64)], 1)    # [b, J]
            J = z1.shape[1]
            AtA = torch.einsum("bcd,bj,bk->cjdk", G, z1, z1).reshape(C * J, C * J)
            Aty = torch.einsum("bcp,bp,bj->cj", U, tg[:, k], z1).reshape(C * J)
            sums[k] = (AtA, Aty) if sums[k] is None else (sums[k][0] + AtA, sums[k][1] + Aty)
    out = {}
    for k in range(3):
        AtA, Aty = sums[k]
        n = AtA.shape[0]
        w = torch.linalg.solve(AtA + lam * AtA.diagonal().mean() * torch.eye(n, dtype=torch.float64), Aty).reshape(C, J)
        out[HYPER.format(k + 1, "weight")] = w[:, :-1].float().contiguous()
        out[HYPER.format(k + 1, "bias")] = w[:, -1].float().contiguous()
    out[IOU_BIAS] = P[IOU_BIAS].clone() + torch.tensor(iou_bias, dtype=P[IOU_BIAS].dtype)
    return out
