"""Generates tests/golden/qwen_*.npz by running the installed transformers (5.15.0) Qwen2.5-VL — the arithmetic
RGA3's UniGRModel inherits (reference model/qwen_2_5_vl_sam2.py:9-12,104) — on tiny deterministic inputs.

Run in the build container only:  python tests/golden/make_qwen_fixtures.py
Weights are NOT stored: they are regenerated from parameter names by oracle/detweights.py.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.detweights import det_state_dict, det_tensor  # noqa: E402

import transformers  # noqa: E402
from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration  # noqa: E402
from transformers.vision_utils import get_vision_cu_seqlens, get_vision_position_ids, get_vision_window_index  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

TINY = dict(
    vision=dict(depth=4, hidden_size=64, num_heads=4, intermediate_size=88, patch_size=14, temporal_patch_size=2,
                spatial_merge_size=2, window_size=112, fullatt_block_indexes=[1, 3], out_hidden_size=96, in_channels=3,
                tokens_per_second=2),
    text=dict(hidden_size=96, num_hidden_layers=2, num_attention_heads=6, num_key_value_heads=2, intermediate_size=160,
              vocab_size=320, rms_norm_eps=1e-6, rope_theta=1000000.0, mrope_section=[2, 3, 3]),
    image_token_id=301, video_token_id=302, vision_start_token_id=303,
)


def hf_name_to_ckpt(n):
    """5.15 module tree -> the 4.49 checkpoint names the reference's released weights use."""
    if n.startswith("model.visual."):
        return n[len("model."):]
    if n.startswith("model.language_model."):
        return "model." + n[len("model.language_model."):]
    return n


def build_hf():
    v, t = TINY["vision"], TINY["text"]
    cfg = Qwen2_5_VLConfig(
        vision_config=dict(depth=v["depth"], hidden_size=v["hidden_size"], num_heads=v["num_heads"],
                           intermediate_size=v["intermediate_size"], patch_size=v["patch_size"],
                           temporal_patch_size=v["temporal_patch_size"], spatial_merge_size=v["spatial_merge_size"],
                           window_size=v["window_size"], fullatt_block_indexes=v["fullatt_block_indexes"],
                           out_hidden_size=v["out_hidden_size"], in_channels=3, tokens_per_second=v["tokens_per_second"],
                           hidden_act="silu"),
        text_config=dict(hidden_size=t["hidden_size"], num_hidden_layers=t["num_hidden_layers"],
                         num_attention_heads=t["num_attention_heads"], num_key_value_heads=t["num_key_value_heads"],
                         intermediate_size=t["intermediate_size"], vocab_size=t["vocab_size"], rms_norm_eps=t["rms_norm_eps"],
                         rope_parameters={"rope_type": "default", "rope_theta": t["rope_theta"], "mrope_section": t["mrope_section"]},
                         max_position_embeddings=4096, tie_word_embeddings=False, hidden_act="silu"),
        image_token_id=TINY["image_token_id"], video_token_id=TINY["video_token_id"],
        vision_start_token_id=TINY["vision_start_token_id"], tie_word_embeddings=False,
    )
    cfg._attn_implementation = "eager"
    cfg.vision_config._attn_implementation = "eager"
    cfg.text_config._attn_implementation = "eager"
    model = Qwen2_5_VLForConditionalGeneration(cfg).float().eval()
    shapes = {hf_name_to_ckpt(n): tuple(p.shape) for n, p in model.named_parameters()}
    sd = det_state_dict(shapes, seed=1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(sd[hf_name_to_ckpt(n)])
    return model, shapes


def seq_with_video(n_vid_tokens, n_pre, n_post, seed):
    g = np.random.default_rng(seed)
    pre = g.integers(0, 300, n_pre)
    post = g.integers(0, 300, n_post)
    return np.concatenate([pre, [TINY["vision_start_token_id"]], np.full(n_vid_tokens, TINY["video_token_id"]), post]).astype(np.int64)


def main():
    torch.manual_seed(0)
    model, shapes = build_hf()
    out = {"transformers_version": np.array(transformers.__version__), "param_names": np.array(sorted(shapes)),
           "param_shapes": np.array([str(shapes[k]) for k in sorted(shapes)])}

    # ---- G4: integer plumbing on several grids (exact / partial windows, odd sizes, full-size config)
    grids = {"a": [[2, 8, 12]], "b": [[3, 18, 26]], "c": [[1, 4, 4], [2, 6, 10]], "full16": [[8, 32, 32]]}
    for key, g in grids.items():
        gt = torch.tensor(g)
        wi, cw = get_vision_window_index(gt, 2, 112, 14)
        out[f"g4_{key}_grid"] = np.array(g)
        out[f"g4_{key}_window_index"] = wi.numpy()
        out[f"g4_{key}_cu_window"] = cw.numpy()
        out[f"g4_{key}_cu_full"] = get_vision_cu_seqlens(gt).numpy()
        out[f"g4_{key}_pos_ids"] = get_vision_position_ids(gt, 2).numpy()

    # rope index: batch of 2 with left/right padding, one video each, different second_per_grid_ts
    for key, (g, spg) in {"a": ([[2, 8, 12]], [1.0]), "b": ([[3, 18, 26]], [2.0])}.items():
        t, h, w = g[0]
        nv = t * (h // 2) * (w // 2)
        ids = seq_with_video(nv, 5, 9, seed=3)
        S = len(ids) + 4
        batch = np.zeros((2, S), dtype=np.int64)
        am = np.zeros((2, S), dtype=np.int64)
        batch[0, :len(ids)] = ids; am[0, :len(ids)] = 1                 # right padded
        batch[1, 4:] = ids; am[1, 4:] = 1                               # left padded
        tt = np.where(batch == TINY["video_token_id"], 2, 0) * am
        pos, delta = model.model.get_rope_index(torch.from_numpy(batch), mm_token_type_ids=torch.from_numpy(tt).int(),
                                                video_grid_thw=torch.tensor(g * 2),
                                                second_per_grid_ts=torch.tensor(spg * 2), attention_mask=torch.from_numpy(am))
        out[f"rope_{key}_input_ids"] = batch
        out[f"rope_{key}_attention_mask"] = am
        out[f"rope_{key}_grid"] = np.array(g * 2)
        out[f"rope_{key}_spg"] = np.array(spg * 2, dtype=np.float32)
        out[f"rope_{key}_position_ids"] = pos.numpy()
        out[f"rope_{key}_deltas"] = delta.numpy()

    # ---- G5: ViT forward on grid a and b
    for key in ("a", "b"):
        g = grids[key]
        n = int(np.prod(g[0]))
        px = det_tensor(f"pixel_values_{key}", (n, 1176), 1.0, seed=5)
        with torch.no_grad():
            vo = model.model.visual(px, grid_thw=torch.tensor(g))
        out[f"vit_{key}_pooler"] = vo.pooler_output.numpy()
        out[f"vit_{key}_last_hidden"] = vo.last_hidden_state.numpy()

    # ---- G5: full forward (video + text), batch 2 with padding, labels -> loss, logits, hidden
    g = [[2, 8, 12]]
    nv = 2 * 4 * 6
    ids = seq_with_video(nv, 6, 10, seed=9)
    S = len(ids) + 3
    batch = np.zeros((2, S), dtype=np.int64); am = np.zeros((2, S), dtype=np.int64)
    batch[0, :len(ids)] = ids; am[0, :len(ids)] = 1
    batch[1, 3:] = ids[::1]; am[1, 3:] = 1
    labels = np.where(am == 1, batch, -100)
    labels[:, : len(ids) - 8] = -100
    labels[1, :] = np.where(np.arange(S) >= S - 6, batch[1], -100)
    px = torch.cat([det_tensor("pixel_values_full0", (2 * 8 * 12, 1176), 1.0, seed=5),
                    det_tensor("pixel_values_full1", (2 * 8 * 12, 1176), 1.0, seed=6)], 0)
    tt = np.where(batch == TINY["video_token_id"], 2, 0) * am
    gt = torch.tensor(g * 2)
    spg = torch.tensor([1.0, 1.0])
    pos, _ = model.model.get_rope_index(torch.from_numpy(batch), mm_token_type_ids=torch.from_numpy(tt).int(), video_grid_thw=gt,
                                        second_per_grid_ts=spg, attention_mask=torch.from_numpy(am))
    with torch.no_grad():
        o = model(input_ids=torch.from_numpy(batch), attention_mask=torch.from_numpy(am), position_ids=pos,
                  labels=torch.from_numpy(labels), pixel_values_videos=px, video_grid_thw=gt, second_per_grid_ts=spg,
                  output_hidden_states=True)
    out["full_input_ids"] = batch
    out["full_attention_mask"] = am
    out["full_labels"] = labels
    out["full_grid"] = np.array(g * 2)
    out["full_position_ids"] = pos.numpy()
    out["full_logits"] = o.logits.numpy()
    out["full_loss"] = o.loss.numpy()
    out["full_hidden_last"] = o.hidden_states[-1].numpy()

    # ---- greedy generate 6 tokens from sample 0 (unpadded) — token ids are a bit-exact target
    ids0 = torch.from_numpy(ids[None])
    tt0 = torch.from_numpy(np.where(ids[None] == TINY["video_token_id"], 2, 0)).int()
    with torch.no_grad():
        gen = model.generate(input_ids=ids0, attention_mask=torch.ones_like(ids0), mm_token_type_ids=tt0,
                             pixel_values_videos=px[: 2 * 8 * 12], video_grid_thw=torch.tensor(g),
                             second_per_grid_ts=torch.tensor([1.0]), max_new_tokens=6, do_sample=False)
    out["gen_input_ids"] = ids[None]
    out["gen_output_ids"] = gen.numpy()

    np.savez_compressed(os.path.join(OUT, "qwen_tiny.npz"), **out)
    print("wrote qwen_tiny.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 0 and not k.startswith("param")})


if __name__ == "__main__":
    main()
