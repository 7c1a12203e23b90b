"""Generates tests/golden/sam2_tiny.npz by importing the REFERENCE's own /root/reference/model/sam2.py (and the loss /
metric helpers of model/qwen_2_5_vl_sam2.py, utils/utils.py, evaluation/mevis_val_u/metrics.py) and running its
classes at tiny sizes on deterministic inputs.  Build container only:  python tests/golden/make_sam2_fixtures.py

Harness shims: qwen_vl_utils stub; Tensor.cuda no-op (sam2.py:2897,2900,3535,3860 hard-code
.cuda()); inference_state device fields forced to cpu after init_state (sam2.py:3790-3791).
Weights are regenerated from parameter names (oracle/detweights.py), never stored.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.modules["qwen_vl_utils"] = types.SimpleNamespace(process_vision_info=lambda *a, **k: None)
torch.Tensor.cuda = lambda self, *a, **k: self
import transformers  # noqa: E402,F401  (resolve its optional-dependency probes before the stubs below exist)
from transformers import LogitsProcessor, Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration  # noqa: E402,F401
# utils/utils.py imports torchvision / matplotlib at module scope only for helpers this harness never calls
for _m in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "matplotlib", "matplotlib.pyplot"):
    if _m not in sys.modules:
        try:
            __import__(_m)
        except Exception:
            sys.modules[_m] = types.SimpleNamespace(resize=None, to_pil_image=None, transforms=None, functional=None, pyplot=None)

from oracle.detweights import det_state_dict, det_tensor  # noqa: E402
from tests.blob_inputs import object_video  # noqa: E402

import model.sam2 as RS  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))

TINY = dict(image_size=128, embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
            pos_bkg=(7, 7), d_model=256, mem_dim=64, memattn_layers=2, memattn_ff=64)


def build_tiny_predictor(image_size=None, overrides=None):
    """overrides: {name: tensor} poured over the name-derived weights (the fitted mask-head read-out, tests/golden/blobfit.py)."""
    t = dict(TINY)
    if image_size is not None:
        t["image_size"] = image_size
    trunk = RS.Hiera(embed_dim=t["embed_dim"], num_heads=t["num_heads"], stages=t["stages"], global_att_blocks=t["global_att_blocks"],
                     window_pos_embed_bkg_spatial_size=t["pos_bkg"], window_spec=t["window_spec"])
    neck = RS.FpnNeck(d_model=t["d_model"], position_encoding=RS.PositionEmbeddingSine(num_pos_feats=t["d_model"], normalize=True, scale=None, temperature=10000),
                      backbone_channel_list=[128, 64, 32, 16], fpn_top_down_levels=[2, 3], fpn_interp_model="nearest")
    enc = RS.ImageEncoder(scalp=1, trunk=trunk, neck=neck)

    def layer():
        sa = RS.RoPEAttention(rope_theta=10000.0, feat_sizes=[8, 8], embedding_dim=t["d_model"], num_heads=1, downsample_rate=1, dropout=0.1)
        ca = RS.RoPEAttention(rope_theta=10000.0, feat_sizes=[8, 8], rope_k_repeat=True, embedding_dim=t["d_model"], num_heads=1,
                              downsample_rate=1, dropout=0.1, kv_in_dim=t["mem_dim"])
        return RS.MemoryAttentionLayer(activation="relu", dim_feedforward=t["memattn_ff"], dropout=0.1, pos_enc_at_attn=False, d_model=t["d_model"],
                                       pos_enc_at_cross_attn_queries=False, pos_enc_at_cross_attn_keys=True, self_attention=sa, cross_attention=ca)

    memattn = RS.MemoryAttention(d_model=t["d_model"], pos_enc_at_input=True, num_layers=t["memattn_layers"], layer=layer())
    memenc = RS.MemoryEncoder(out_dim=t["mem_dim"], position_encoding=RS.PositionEmbeddingSine(num_pos_feats=t["mem_dim"], normalize=True, scale=None, temperature=10000),
                              mask_downsampler=RS.MaskDownSampler(embed_dim=t["d_model"], kernel_size=3, stride=2, padding=1),
                              fuser=RS.Fuser(layer=RS.CXBlock(dim=t["d_model"], kernel_size=7, padding=3, layer_scale_init_value=1e-6, use_dwconv=True), num_layers=2),
                              in_dim=t["d_model"])
    pred = RS.SAM2VideoPredictor(
        image_encoder=enc, memory_attention=memattn, memory_encoder=memenc, num_maskmem=7, image_size=t["image_size"],
        sigmoid_scale_for_mem_enc=20.0, sigmoid_bias_for_mem_enc=-10.0, use_mask_input_as_output_without_sam=True, directly_add_no_mem_embed=True,
        use_high_res_features_in_sam=True, multimask_output_in_sam=True, iou_prediction_use_sigmoid=True, use_obj_ptrs_in_encoder=True,
        add_tpos_enc_to_obj_ptrs=False, only_obj_ptrs_in_the_past_for_eval=True, pred_obj_scores=True, pred_obj_scores_mlp=True, fixed_no_obj_ptr=True,
        multimask_output_for_tracking=True, use_multimask_token_for_obj_ptr=True, multimask_min_pt_num=0, multimask_max_pt_num=1,
        use_mlp_for_obj_ptr_proj=True, compile_image_encoder=False,
        sam_mask_decoder_extra_args={"dynamic_multimask_via_stability": True, "dynamic_multimask_stability_delta": 0.05, "dynamic_multimask_stability_thresh": 0.98})
    pred = pred.float().eval()
    shapes = {k: tuple(v.shape) for k, v in pred.state_dict().items()}
    sd = det_state_dict(shapes, seed=2)
    sd.update(overrides or {})
    sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}   # bf16-representable weights: reference (fp32 compute), oracle and product see identical numbers
    pred.load_state_dict(sd, strict=True)
    # the SAM2 wrapper hard-codes SAM2-L sizes in its constructor (sam2.py:87-146); wrap the tiny predictor in it without running it
    wrap = RS.SAM2.__new__(RS.SAM2)
    torch.nn.Module.__init__(wrap)
    wrap.sam2_model = pred
    wrap.hidden_dim = pred.hidden_dim
    return wrap, shapes


def count_calls(pred):
    counts = {"enc": 0, "memattn": 0, "memenc": 0, "dec": 0}
    def hook(name):
        def f(*a, **k):
            counts[name] += 1
        return f
    hs = [pred.image_encoder.register_forward_pre_hook(hook("enc")), pred.memory_attention.register_forward_pre_hook(hook("memattn")),
          pred.memory_encoder.register_forward_pre_hook(hook("memenc")), pred.sam_mask_decoder.register_forward_pre_hook(hook("dec"))]
    return counts, hs


def main():
    out = {}
    # ------------------------------------------------------------------ G1: pure functions
    cis = RS.compute_axial_cis(dim=32, end_x=4, end_y=4)
    out["g1_axial_cos"], out["g1_axial_sin"] = cis.real.numpy(), cis.imag.numpy()
    xq, xk = det_tensor("g1_xq", (1, 1, 16, 32)), det_tensor("g1_xk", (1, 1, 37, 32))
    q2, k2 = RS.apply_rotary_enc(xq, xk[:, :, :32].clone(), freqs_cis=cis, repeat_freqs_k=True)
    out["g1_rot_q"], out["g1_rot_k"] = q2.numpy(), k2.numpy()
    xw = det_tensor("g1_win", (2, 10, 13, 6))
    win, pad = RS.window_partition(xw, 4)
    out["g1_win_part"], out["g1_win_pad"] = win.numpy(), np.array(pad)
    out["g1_win_unpart"] = RS.window_unpartition(win, 4, pad, (10, 13)).numpy()
    pes = RS.PositionEmbeddingSine(num_pos_feats=32)
    out["g1_pe_sine"] = pes(torch.zeros(2, 5, 6, 7))[0].numpy()
    per = RS.PositionEmbeddingRandom(16)
    G = det_tensor("g1_gauss", (2, 16))
    per.positional_encoding_gaussian_matrix = G
    out["g1_pe_random"] = per((5, 7)).numpy()
    out["g1_1d_sine"] = RS.get_1d_sine_pe(torch.tensor([0.0, 0.25, 1.0]), dim=16).numpy()
    cond = {t: {"t": t} for t in (0, 3, 9, 14, 20)}
    sel, unsel = RS.select_closest_cond_frames(10, cond, 3)
    out["g1_sel_cond"], out["g1_unsel_cond"] = np.array(sorted(sel)), np.array(sorted(unsel))
    ln2 = RS.LayerNorm2d(6)
    ln2.weight.data, ln2.bias.data = det_tensor("g1_ln2w", (6,), 0.5, offset=1.0), det_tensor("g1_ln2b", (6,), 0.2)
    out["g1_ln2d"] = ln2(det_tensor("g1_ln2x", (2, 6, 3, 4))).detach().numpy()

    # losses / metrics from the reference's own helpers
    from model.qwen_2_5_vl_sam2 import dice_loss, sigmoid_ce_loss
    pm, gm = det_tensor("g1_pred_mask", (3, 9, 11), 3.0), (det_tensor("g1_gt_mask", (3, 9, 11)) > 0.3).float()
    out["g1_dice"], out["g1_bce"] = dice_loss(pm, gm, 3).numpy(), sigmoid_ce_loss(pm, gm, 3).numpy()
    out["g1_dice_empty"], out["g1_bce_empty"] = dice_loss(pm[0:0], gm[0:0], 0).numpy(), sigmoid_ce_loss(pm[0:0], gm[0:0], 0).numpy()
    from utils.utils import intersectionAndUnionGPU
    a, b = (det_tensor("g1_iou_a", (8, 8)) > 0).long(), (det_tensor("g1_iou_b", (8, 8)) > 0.2).long()
    i_, u_, t_ = intersectionAndUnionGPU(a.clone().float(), b.clone().float(), 2, ignore_index=255)
    out["g1_iau"] = np.stack([i_.numpy(), u_.numpy(), t_.numpy()])
    # db_eval_iou (evaluation/mevis_val_u/metrics.py) is not importable offline (cv2 / skimage at module import): its
    # definition (sum(seg & gt) / sum(seg | gt), empty union -> 1) is restated in the product's metrics and pinned by hand cases.

    # ------------------------------------------------------------------ G2/G3: tiny SAM2 built from the reference classes
    T = 5
    imgs, obj = object_video("sam_images", T, 128, seed=3)     # an ellipse drifting over a smooth background (tests/blob_inputs.py)
    imgs = imgs.to(torch.bfloat16).float()
    emb = det_tensor("lang_embd", (T, 1, 256), 1.0, seed=4).to(torch.bfloat16).float()
    # fit the mask head's read-out to that object on frames 0-2 (frames 3-4 and the memory path are then genuinely predicted)
    import blobfit as BF
    from tests.sam2_tiny import tiny_cfg
    _, shapes = build_tiny_predictor()
    fitted = BF.fit({k: v.to(torch.bfloat16).float() for k, v in det_state_dict(shapes, seed=2).items()}, tiny_cfg(), imgs[:3], emb[:3], obj[:3])
    fitted = {k: v.to(torch.bfloat16).float() for k, v in fitted.items()}
    for k, v in fitted.items():
        out["fit::" + k] = v.numpy()
    wrap, shapes = build_tiny_predictor(overrides=fitted)
    pred = wrap.sam2_model
    out["param_names"] = np.array(sorted(shapes))
    out["param_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes)])

    with torch.no_grad():
        # G2: image encoder levels
        bo = pred.forward_image(imgs[:2])
        for i, (f, p) in enumerate(zip(bo["backbone_fpn"], bo["vision_pos_enc"])):
            out[f"g2_fpn_{i}"], out[f"g2_pos_{i}"] = f.numpy(), p[0].numpy()
        trunk_out = pred.image_encoder.trunk(imgs[:1])
        for i, f in enumerate(trunk_out):
            out[f"g2_trunk_{i}"] = f.numpy()
        # G3a: training path (frames independent)
        st = wrap.get_sam2_embeddings_train(imgs[:3])
        low, high = wrap.inject_language_embd_train(st, emb[:3])
        out["g3_train_low"], out["g3_train_high"] = low.numpy(), high.numpy()
        # internals of the same call for decoder-level pinning
        feats = st
        hr = [x.permute(1, 2, 0).view(x.size(1), x.size(2), *s) for x, s in zip(feats["current_vision_feats"][:-1], feats["feat_sizes"][:-1])]
        pix = (feats["current_vision_feats"][-1] + pred.no_mem_embed).permute(1, 2, 0).view(3, 256, 8, 8)
        lm, hm, ious, lr, hrm, optr, osl = pred._forward_sam_heads(backbone_features=pix, high_res_features=hr, multimask_output=True, language_embd=emb[:3])
        out["g3_heads_ious"], out["g3_heads_best"] = ious.numpy(), torch.argmax(ious, -1).numpy()
        out["g3_heads_low_multi"], out["g3_heads_obj_ptr"], out["g3_heads_obj_logits"] = lm.numpy(), optr.numpy(), osl.numpy()
        # G2: memory encoder on the chosen high-res masks
        mf, mp = pred._encode_new_memory(feats["current_vision_feats"], feats["feat_sizes"], hrm, False)
        out["g2_memenc_feat"], out["g2_memenc_pos"] = mf.numpy(), mp[0].numpy()
        # G2: memory attention on synthetic memory (two 8x8 frames of mem + 4 pointer tokens)
        mem = det_tensor("g2_mem", (2 * 64 + 8, 1, 64)); mem_pos = det_tensor("g2_mem_pos", (2 * 64 + 8, 1, 64))
        cur = feats["current_vision_feats"][-1][:, :1]; cur_pos = feats["current_vision_pos_embeds"][-1][:, :1]
        ma = pred.memory_attention(curr=[cur], curr_pos=[cur_pos], memory=mem, memory_pos=mem_pos, num_obj_ptr_tokens=8)
        out["g2_memattn"] = ma.numpy()

        # G3b: reference usage — prompt on every frame (sam2.py:378-404)
        counts, hs = count_calls(pred)
        state = wrap.get_sam2_embeddings(imgs)
        state["device"] = state["storage_device"] = torch.device("cpu")
        masks = wrap.language_embd_inference(state, [emb[t] for t in range(T)])
        out["g3_infer_all_masks"] = masks.numpy()
        out["g3_infer_all_counts"] = np.array([counts[k] for k in ("enc", "memattn", "memenc", "dec")])
        for h in hs:
            h.remove()

        # G3c: prompt on frame 0 only, then propagate (memory attention active)
        counts, hs = count_calls(pred)
        state = pred.init_state(imgs)
        state["device"] = state["storage_device"] = torch.device("cpu")
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            pred.add_language_embd(state, 0, 100, emb[0][None], inference=True)
            res = [m for _, _, m in pred.propagate_in_video(state)]
        out["g3_prop0_masks"] = torch.cat(res, 0).numpy()
        out["g3_prop0_counts"] = np.array([counts[k] for k in ("enc", "memattn", "memenc", "dec")])
        out["g3_prop0_obj_ptrs"] = np.stack([state["output_dict"]["cond_frame_outputs" if t == 0 else "non_cond_frame_outputs"][t]["obj_ptr"].numpy() for t in range(T)])
        for h in hs:
            h.remove()

    # margins of what was just pinned (printed for the record: a threshold test needs |logit| well away from 0 on most pixels)
    for key in ("g3_train_high", "g3_infer_all_masks", "g3_prop0_masks"):
        m = torch.from_numpy(out[key]).reshape(-1, 128, 128)
        print(key, "margin(|x| > 5% max) per frame:", [round(float((x.abs() > 0.05 * x.abs().max()).float().mean()), 3) for x in m],
              "IoU with the object:", [round(float(((x > 0) & o).sum() / ((x > 0) | o).sum().clamp(min=1)), 3) for x, o in zip(m, obj)])
    np.savez_compressed(os.path.join(OUT, "sam2_tiny.npz"), **out)
    print("wrote sam2_tiny.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("param")})


if __name__ == "__main__":
    main()
