"""Generates tests/golden/unigr_tiny.npz from the REFERENCE's UniGRModel (model/qwen_2_5_vl_sam2.py) built on a tiny
transformers-5.15 Qwen2.5-VL and a tiny SAM2 assembled from the reference's own sam2.py classes.
Build container only:  python tests/golden/make_unigr_fixtures.py     (shims: SURVEY.md Appendix D)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_sam2_fixtures as MS  # noqa: E402  (installs the shims, imports the reference)
import make_qwen_fixtures as MQ  # noqa: E402
from oracle.detweights import det_state_dict, det_tensor  # noqa: E402

import model.qwen_2_5_vl_sam2 as RU  # noqa: E402  (the reference)

SEG = 300
T_SAM = 2
SAM_SIDE = 1024  # reference model_forward hard-codes 256x256 / 1024x1024 mask sizes (:267-268)


def build():
    v, t = MQ.TINY["vision"], MQ.TINY["text"]
    cfg = RU.UniGRConfig(
        train_mask_decoder=True, out_dim=256, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seg_token_idx=SEG, sam_pretrained=None,
        vision_config=dict(depth=v["depth"], hidden_size=v["hidden_size"], num_heads=v["num_heads"], intermediate_size=v["intermediate_size"],
                           patch_size=14, temporal_patch_size=2, spatial_merge_size=2, window_size=112, fullatt_block_indexes=v["fullatt_block_indexes"],
                           out_hidden_size=v["out_hidden_size"], in_channels=3, tokens_per_second=2, hidden_act="silu"),
        text_config=dict(hidden_size=t["hidden_size"], num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
                         num_key_value_heads=t["num_key_value_heads"], intermediate_size=t["intermediate_size"], vocab_size=t["vocab_size"],
                         rms_norm_eps=1e-6, rope_parameters={"rope_type": "default", "rope_theta": t["rope_theta"], "mrope_section": t["mrope_section"]},
                         max_position_embeddings=4096, tie_word_embeddings=False, hidden_act="silu"),
        image_token_id=301, video_token_id=302, vision_start_token_id=303, tie_word_embeddings=False)
    cfg.hidden_size = cfg.text_config.hidden_size  # reference reads config.hidden_size (:129); 5.x nests it
    cfg._attn_implementation = "eager"
    cfg.vision_config._attn_implementation = "eager"
    cfg.text_config._attn_implementation = "eager"
    model = RU.UniGRModel(cfg).float()
    wrap, sam_shapes = MS.build_tiny_predictor(SAM_SIDE)
    RU.SAM2 = lambda ckpt_path=None: wrap       # initialize_sam_modules builds SAM2-L otherwise (reference :119)
    model.initialize_sam_modules(cfg)
    model = model.float()
    names = {}
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("grounding_encoder."):
                continue
            ck = MQ.hf_name_to_ckpt(n)
            names[ck] = tuple(p.shape)
        sd = det_state_dict(names, seed=1)
        for n, p in model.named_parameters():
            if not n.startswith("grounding_encoder."):
                p.copy_(sd[MQ.hf_name_to_ckpt(n)])
    return model, names, sam_shapes


def make_batch(seg_flags, seed):
    g = np.random.default_rng(seed)
    grid = [[2, 8, 12]]
    nv = 2 * 4 * 6
    ids, labs = [], []
    for b, has in enumerate(seg_flags):
        pre = g.integers(0, 290, 6)
        ans = g.integers(0, 290, 7)
        if has:
            ans[3] = SEG
        seq = np.concatenate([pre, [303], np.full(nv, 302), g.integers(0, 290, 4), ans]).astype(np.int64)
        lab = np.full_like(seq, -100)
        lab[-7:] = seq[-7:]
        ids.append(seq); labs.append(lab)
    ids, labs = np.stack(ids), np.stack(labs)
    B = len(seg_flags)
    px = torch.cat([det_tensor(f"unigr_px_{seed}_{b}", (2 * 8 * 12, 1176), 1.0, seed=5) for b in range(B)], 0)
    imgs = torch.stack([det_tensor(f"unigr_img_{seed}_{b}", (T_SAM, 3, SAM_SIDE, SAM_SIDE), 1.0, seed=6) for b in range(B)], 0)
    h, w = 20, 28
    masks = []
    for b, has in enumerate(seg_flags):
        m = (det_tensor(f"unigr_gt_{seed}_{b}", (T_SAM, h, w), 1.0, seed=7) > 0.3).float()
        masks.append(m if has else m[0:0])
    return dict(input_ids=torch.from_numpy(ids), labels=torch.from_numpy(labs), attention_mask=torch.ones(B, ids.shape[1], dtype=torch.long),
                pixel_values_videos=px, video_grid_thw=torch.tensor(grid * B), second_per_grid_ts=torch.tensor([1.0] * B), images_sam=imgs,
                offset=torch.arange(B + 1), masks_list=masks, label_list=[torch.zeros(h, w) for _ in range(B)], resize_list=[(SAM_SIDE, SAM_SIDE)] * B)


def main():
    model, names, sam_shapes = build()
    out = {"param_names": np.array(sorted(names)), "param_shapes": np.array([str(names[k]) for k in sorted(names)]),
           "sam_param_names": np.array(sorted(sam_shapes)), "sam_param_shapes": np.array([str(sam_shapes[k]) for k in sorted(sam_shapes)])}
    model.train()
    model.grounding_encoder.sam2_model.eval()
    model.grounding_encoder.sam2_model.sam_mask_decoder.train()
    for case, flags in {"11": (True, True), "10": (True, False), "00": (False, False)}.items():
        b = make_batch(flags, seed=int(case, 2) + 1)
        tt = (b["input_ids"] == 302).int() * 2
        pos, _ = model.model.get_rope_index(b["input_ids"], mm_token_type_ids=tt, video_grid_thw=b["video_grid_thw"],
                                            second_per_grid_ts=b["second_per_grid_ts"], attention_mask=b["attention_mask"])
        model.zero_grad()
        o = model(**b, position_ids=pos, inference=False)
        for k in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss"):
            out[f"train_{case}_{k}"] = np.float64(float(o[k]))
        o["loss"].backward()
        g = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        for n in ("text_hidden_fcs.0.2.weight", "lm_head.weight", "grounding_encoder.sam2_model.sam_mask_decoder.output_hypernetworks_mlps.1.layers.2.weight",
                  "grounding_encoder.sam2_model.sam_mask_decoder.transformer.layers.0.cross_attn_token_to_image.q_proj.weight"):
            if n in g:
                out[f"train_{case}_grad::{n}"] = g[n].numpy().copy()
        out[f"train_{case}_n_grads"] = np.int64(len(g))
        out[f"train_{case}_input_ids"] = b["input_ids"].numpy()
        out[f"train_{case}_labels"] = b["labels"].numpy()

    # evaluate(): teacher-forced "Sure, [SEG]." in input_ids, one sample
    model.eval()
    b = make_batch((True,), seed=9)
    tt = (b["input_ids"] == 302).int() * 2
    pos, _ = model.model.get_rope_index(b["input_ids"], mm_token_type_ids=tt, video_grid_thw=b["video_grid_thw"],
                                        second_per_grid_ts=b["second_per_grid_ts"], attention_mask=b["attention_mask"])
    # evaluate() does not forward position_ids (:346-355): under transformers 5.x that silently means 1-D positions, so the
    # harness routes the explicit ids through a patched super().forward kwarg (same arithmetic as 4.49's internal get_rope_index)
    import transformers
    orig = transformers.Qwen2_5_VLForConditionalGeneration.forward
    def fwd(self, *a, **k):
        k.setdefault("position_ids", pos)
        if k.get("position_ids") is None:
            k["position_ids"] = pos
        return orig(self, *a, **k)
    transformers.Qwen2_5_VLForConditionalGeneration.forward = fwd
    state_patch = model.grounding_encoder.sam2_model.init_state
    def init_state(images):
        st = state_patch(images)
        st["device"] = st["storage_device"] = torch.device("cpu")
        return st
    model.grounding_encoder.sam2_model.init_state = init_state
    with torch.no_grad():
        o, masks = model.evaluate(b["input_ids"], b["attention_mask"], None, b["pixel_values_videos"], None, b["video_grid_thw"], b["second_per_grid_ts"],
                                  b["images_sam"], b["resize_list"], [(20, 28)])
    transformers.Qwen2_5_VLForConditionalGeneration.forward = orig
    out["eval_input_ids"] = b["input_ids"].numpy()
    out["eval_masks"] = masks[0].numpy()
    out["eval_n_masks"] = np.int64(len(masks))
    np.savez_compressed(os.path.join(HERE, "unigr_tiny.npz"), **out)
    print("wrote unigr_tiny.npz", {k: (v.shape if hasattr(v, "shape") and v.ndim else v) for k, v in out.items() if "param" not in k and "grad::" not in k})


if __name__ == "__main__":
    main()
