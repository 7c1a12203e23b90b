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
    _, sam_shapes = MS.build_tiny_predictor(SAM_SIDE)
    # ---- names / deterministic weights of the Qwen side + text_hidden_fcs (created by initialize_sam_modules: built once with a throw-away SAM2)
    RU.SAM2 = lambda ckpt_path=None: MS.build_tiny_predictor(SAM_SIDE)[0]
    model.initialize_sam_modules(cfg)
    names = {MQ.hf_name_to_ckpt(n): tuple(p.shape) for n, p in model.named_parameters() if not n.startswith("grounding_encoder.")}
    sd = {k: v.to(torch.bfloat16).float() for k, v in det_state_dict(names, seed=1).items()}   # bf16-representable, like every weight of these fixtures
    # ---- fit the mask head's read-out to the objects in the clips of the [SEG] samples of cases "11" and "10" (blobfit.py); the language embeddings
    #      are the ones the (oracle) LLM side produces for those samples; the evaluate / inference clips are NOT part of the fit
    import blobfit as BF
    from oracle import unigr as OU
    PS0 = {k: v.to(torch.bfloat16).float() for k, v in det_state_dict(sam_shapes, seed=2).items()}
    fit_imgs, fit_emb, fit_obj = [], [], []
    for case in ("11", "10"):
        flags = CASES[case]
        b = make_batch(flags, seed=int(case, 2) + 1)
        with torch.no_grad():
            r = OU.model_forward(sd, PS0, oracle_cfg(), sam_cfg(), b, (1.0, 0.5, 2.0), SEG)
        objs = object_masks(flags, seed=int(case, 2) + 1)
        k = 0
        for i, has in enumerate(flags):
            if has:
                fit_imgs.append(b["images_sam"][i]); fit_obj.append(objs[i]); fit_emb.append(r["pred_embeddings"][k:k + 1][None].expand(T_SAM, 1, -1)); k += 1
    fitted = BF.fit(PS0, sam_cfg(), torch.cat(fit_imgs), torch.cat(fit_emb), torch.cat(fit_obj), chunk=2)
    fitted = {k: v.to(torch.bfloat16).float() for k, v in fitted.items()}
    wrap, sam_shapes = MS.build_tiny_predictor(SAM_SIDE, overrides=fitted)
    RU.SAM2 = lambda ckpt_path=None: wrap       # initialize_sam_modules builds SAM2-L otherwise (reference :119)
    model.initialize_sam_modules(cfg)
    model = model.float()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if not n.startswith("grounding_encoder."):
                p.copy_(sd[MQ.hf_name_to_ckpt(n)])
    return model, names, sam_shapes, fitted


from tests.unigr_tiny import CASES, LABEL_HW, make_batch, object_masks, sam_cfg  # noqa: E402  (ONE definition of the synthetic batches)
from tests.qwen_tiny import oracle_cfg  # noqa: E402


def main():
    model, names, sam_shapes, fitted = build()
    out = {"param_names": np.array(sorted(names)), "param_shapes": np.array([str(names[k]) for k in sorted(names)]),
           "sam_param_names": np.array(sorted(sam_shapes)), "sam_param_shapes": np.array([str(sam_shapes[k]) for k in sorted(sam_shapes)])}
    for k, v in fitted.items():
        out["fit::" + k] = v.numpy()
    model.train()
    model.grounding_encoder.sam2_model.eval()
    model.grounding_encoder.sam2_model.sam_mask_decoder.train()
    for case, flags in CASES.items():
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
                  "grounding_encoder.sam2_model.sam_mask_decoder.transformer.layers.0.cross_attn_token_to_image.q_proj.weight",
                  "grounding_encoder.sam2_model.sam_mask_decoder.output_upscaling.0.weight", "text_hidden_fcs.0.0.weight"):
            if n in g:
                out[f"train_{case}_grad::{n}"] = g[n].numpy().copy()
        out[f"train_{case}_n_grads"] = np.int64(len(g))
        out[f"train_{case}_input_ids"] = b["input_ids"].numpy()
        out[f"train_{case}_labels"] = b["labels"].numpy()

    # ---- H1: two optimizer steps of the reference's recipe on the "11" batch -- trainable set of train_joint.py:237-251 (no
    #      LoRA here: PEFT is absent), gradient clipping 1.0, AdamW lr 4e-5, betas (0.9, 0.95), wd 0 (train_joint.py:300-324; DeepSpeed's FusedAdam in
    #      AdamW mode == torch.optim.AdamW).  Pinned: loss dict before each step, the pre-clip gradient norm, gradients and parameter deltas of 4 tensors.
    H1_TENSORS = ("lm_head.weight", "model.language_model.embed_tokens.weight", "text_hidden_fcs.0.2.weight",
                  "grounding_encoder.sam2_model.sam_mask_decoder.output_hypernetworks_mlps.1.layers.2.weight")
    saved = {n: p.detach().clone() for n, p in model.named_parameters()}
    flags_before = {n: p.requires_grad for n, p in model.named_parameters()}
    for n, p in model.named_parameters():
        p.requires_grad_(any(x in n for x in ("lm_head", "embed_tokens", "sam_mask_decoder", "text_hidden_fcs")))
    train = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(train, lr=4e-5, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0)
    b = make_batch(CASES["11"], seed=int("11", 2) + 1)
    tt = (b["input_ids"] == 302).int() * 2
    pos, _ = model.model.get_rope_index(b["input_ids"], mm_token_type_ids=tt, video_grid_thw=b["video_grid_thw"],
                                        second_per_grid_ts=b["second_per_grid_ts"], attention_mask=b["attention_mask"])
    named = dict(model.named_parameters())
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        o = model(**b, position_ids=pos, inference=False)
        for k in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss"):
            out[f"h1_step{step}_{k}"] = np.float64(float(o[k]))
        o["loss"].backward()
        out[f"h1_step{step}_grad_norm"] = np.float64(float(torch.nn.utils.clip_grad_norm_(train, 1.0)))
        if step == 0:
            for n in H1_TENSORS:
                out[f"h1_grad::{MQ.hf_name_to_ckpt(n) if not n.startswith(('grounding_encoder', 'text_hidden')) else n}"] = (named[n].grad / min(1.0, 1.0 / (out["h1_step0_grad_norm"] + 1e-6))).numpy().copy()
        opt.step()
        for n in H1_TENSORS:
            key = MQ.hf_name_to_ckpt(n) if not n.startswith(("grounding_encoder", "text_hidden")) else n
            out[f"h1_delta{step + 1}::{key}"] = (named[n].detach() - saved[n]).numpy().copy()
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(saved[n])
            p.requires_grad_(flags_before[n])
    model.zero_grad(set_to_none=True)

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
                                  b["images_sam"], b["resize_list"], [LABEL_HW])
    # ---- model_forward(inference=True): the branch validate() drives (reference :236-257, train_joint.py:586-648), batch size 1, with and without [SEG]
    for tag, flags, seed in (("1", (True,), 11), ("0", (False,), 12)):
        bi = make_batch(flags, seed=seed)
        tti = (bi["input_ids"] == 302).int() * 2
        posi, _ = model.model.get_rope_index(bi["input_ids"], mm_token_type_ids=tti, video_grid_thw=bi["video_grid_thw"],
                                             second_per_grid_ts=bi["second_per_grid_ts"], attention_mask=bi["attention_mask"])
        with torch.no_grad():
            oi = model(**bi, position_ids=posi, inference=True)
        assert set(oi) == {"pred_masks", "gt_masks"} and len(oi["pred_masks"]) == 1
        out[f"infer_{tag}_pred_masks"] = oi["pred_masks"][0].numpy()
        out[f"infer_{tag}_input_ids"] = bi["input_ids"].numpy()
    transformers.Qwen2_5_VLForConditionalGeneration.forward = orig
    out["eval_input_ids"] = b["input_ids"].numpy()
    out["eval_masks"] = masks[0].numpy()
    out["eval_n_masks"] = np.int64(len(masks))
    np.savez_compressed(os.path.join(HERE, "unigr_tiny.npz"), **out)
    print("wrote unigr_tiny.npz", {k: (v.shape if hasattr(v, "shape") and v.ndim else v) for k, v in out.items() if "param" not in k and "grad::" not in k})


if __name__ == "__main__":
    main()
