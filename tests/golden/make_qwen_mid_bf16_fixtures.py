"""Generates tests/golden/qwen_mid_bf16.npz: the installed transformers (5.15.0) Qwen2.5-VL run in **bf16 on the CPU** -- the storage precision the reference runs
the model in (app.py:53-58 torch_dtype=torch.bfloat16; train_joint.py:165-179 precision bf16) -- at a MID size where what bf16 storage does is separable from weight
chaos: 8 ViT blocks (d = 256) + 12 decoder layers (d = 512), conditioned random weights (residual-branch output projections / sqrt(2 L), the rule of
tests/test_fulldepth_parity_gpu.py::_conditioned_init), one 2 x 16 x 16 video + text, S = 176.  Beside it the SAME module in fp32 on the same bf16-rounded weights.

What the fixture pins: `oracle.qwen25vl.storage(torch.bfloat16)` -- the yardstick of the full-depth GPU test -- must reproduce transformers'
own bf16 run, not merely be "an fp32 restatement with roundings somewhere".  tests/test_oracle_qwen.py checks it on the CPU.

Run in the build container only:  python tests/golden/make_qwen_mid_bf16_fixtures.py     (weights are not stored: regenerated from the seed below)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import transformers  # noqa: E402
from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

MID = dict(
    vision=dict(depth=8, hidden_size=256, num_heads=4, intermediate_size=688, patch_size=14, temporal_patch_size=2, spatial_merge_size=2, window_size=112,
                fullatt_block_indexes=[3, 7], out_hidden_size=512, in_channels=3, tokens_per_second=2),
    text=dict(hidden_size=512, num_hidden_layers=12, num_attention_heads=8, num_key_value_heads=2, intermediate_size=1408, vocab_size=2048, rms_norm_eps=1e-6,
              rope_theta=1000000.0, mrope_section=[8, 12, 12]),
    image_token_id=2001, video_token_id=2002, vision_start_token_id=2003, vision_end_token_id=2004,
)
GRID = [[2, 16, 16]]
SEED = 91


def hf_name_to_ckpt(n):
    if n.startswith("model.visual."):
        return n[len("model."):]
    if n.startswith("model.language_model."):
        return "model." + n[len("model.language_model."):]
    return n


def mid_state_dict(shapes, seed=SEED):
    """name -> fp32 tensor already rounded to bf16 (both sides of every comparison hold the same weights).  Sorted-name order, one generator: reproducible from the
    shapes alone (tests/qwen_mid.py rebuilds it for the oracle)."""
    Lv, Lt = MID["vision"]["depth"], MID["text"]["num_hidden_layers"]
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n in sorted(shapes):
        shp = tuple(shapes[n])
        if n == "model.embed_tokens.weight":
            t = torch.randn(shp, generator=g)
        elif len(shp) >= 2:
            t = torch.randn(shp, generator=g) * 0.02
            if n.startswith("visual.blocks.") and n.endswith(("attn.proj.weight", "mlp.down_proj.weight")):
                t = t * (2 * Lv) ** -0.5
            elif n.startswith("model.layers.") and n.endswith(("self_attn.o_proj.weight", "mlp.down_proj.weight")):
                t = t * (2 * Lt) ** -0.5
        elif "norm" in n or "ln_q" in n:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g) * 0.02
        sd[n] = t.to(torch.bfloat16).float()
    return sd


def build_hf(attn):
    v, t = MID["vision"], MID["text"]
    cfg = Qwen2_5_VLConfig(
        vision_config=dict(depth=v["depth"], hidden_size=v["hidden_size"], num_heads=v["num_heads"], intermediate_size=v["intermediate_size"],
                           patch_size=v["patch_size"], temporal_patch_size=v["temporal_patch_size"], spatial_merge_size=v["spatial_merge_size"],
                           window_size=v["window_size"], fullatt_block_indexes=v["fullatt_block_indexes"], out_hidden_size=v["out_hidden_size"], in_channels=3,
                           tokens_per_second=v["tokens_per_second"], hidden_act="silu"),
        text_config=dict(hidden_size=t["hidden_size"], num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
                         num_key_value_heads=t["num_key_value_heads"], intermediate_size=t["intermediate_size"], vocab_size=t["vocab_size"],
                         rms_norm_eps=t["rms_norm_eps"], rope_parameters={"rope_type": "default", "rope_theta": t["rope_theta"], "mrope_section": t["mrope_section"]},
                         max_position_embeddings=4096, tie_word_embeddings=False, hidden_act="silu"),
        image_token_id=MID["image_token_id"], video_token_id=MID["video_token_id"], vision_start_token_id=MID["vision_start_token_id"],
        vision_end_token_id=MID["vision_end_token_id"], tie_word_embeddings=False,
    )
    for c in (cfg, cfg.vision_config, cfg.text_config):
        c._attn_implementation = attn
    model = Qwen2_5_VLForConditionalGeneration(cfg).float().eval()
    shapes = {hf_name_to_ckpt(n): tuple(p.shape) for n, p in model.named_parameters()}
    sd = mid_state_dict(shapes)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(sd[hf_name_to_ckpt(n)])
    return model, shapes


def inputs():
    g = torch.Generator().manual_seed(SEED + 1)
    t, h, w = GRID[0]
    nv = t * (h // 2) * (w // 2)
    px = torch.randn(t * h * w, 1176, generator=g).clamp_(-1.8, 2.2).to(torch.bfloat16).float()
    text = torch.randint(0, 2000, (46,), generator=g)
    ids = torch.cat([text[:14], torch.tensor([MID["vision_start_token_id"]]), torch.full((nv,), MID["video_token_id"]), torch.tensor([MID["vision_end_token_id"]]),
                     text[14:]])[None]
    return px, ids, torch.ones_like(ids)


VIT_PINS, LLM_PINS = (0, 3, 7), (0, 5, 11)          # blocks / layers whose (input, output) pair of the bf16 run is stored: windowed, full-attention, last


def run(model, px, ids, am, dtype):
    m = model.to(dtype)
    gt = torch.tensor(GRID)
    tt = (ids == MID["video_token_id"]).int() * 2
    spg = torch.tensor([1.0])
    pos, _ = m.model.get_rope_index(ids, mm_token_type_ids=tt, video_grid_thw=gt, second_per_grid_ts=spg, attention_mask=am)
    caps, hooks = {}, []

    def grab(key, which):
        def f(mod, args, kwargs, out):
            caps[key + "_in"] = (args[0] if args else kwargs["hidden_states"]).detach().float().reshape(-1, (args[0] if args else kwargs["hidden_states"]).shape[-1])
            o = out[0] if isinstance(out, tuple) else out
            caps[key + "_out"] = o.detach().float().reshape(-1, o.shape[-1])
        return f

    for k in VIT_PINS:
        hooks.append(m.model.visual.blocks[k].register_forward_hook(grab(f"vit{k}", k), with_kwargs=True))
    for k in LLM_PINS:
        hooks.append(m.model.language_model.layers[k].register_forward_hook(grab(f"llm{k}", k), with_kwargs=True))
    with torch.no_grad():
        o = m(input_ids=ids, attention_mask=am, position_ids=pos, pixel_values_videos=px.to(dtype), video_grid_thw=gt, second_per_grid_ts=spg, output_hidden_states=True)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        vo = m.model.visual(px.to(dtype), grid_thw=gt)
    hs = [h[0].float() for h in o.hidden_states]        # embeddings + the residual stream after every decoder layer (the last one post-norm in 5.x)
    return vo.pooler_output.float(), hs, o.logits[0].float(), pos, caps


def bits(t):
    """bf16 values (held in f32) -> their 16 bits: exact, half the bytes"""
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    px, ids, am = inputs()
    out = {"transformers_version": np.array(transformers.__version__), "torch_version": np.array(torch.__version__), "grid": np.array(GRID), "input_ids": ids.numpy(),
           "seed": np.array(SEED), "attn_implementation": np.array("sdpa"), "vit_pins": np.array(VIT_PINS), "llm_pins": np.array(LLM_PINS)}
    # sdpa: torch's fused CPU attention rounds like the flash kernels the reference runs on its GPUs (un-normalised exponentials in bf16 for the second product); the
    # eager form (scores and normalised probabilities rounded) is 4 x further from it on the attention output and is not what the reference executes (train_joint.py:181)
    model, shapes = build_hf("sdpa")
    v32, h32, l32, pos, _ = run(model, px, ids, am, torch.float32)
    v16, h16, l16, _, caps = run(model, px, ids, am, torch.bfloat16)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print("bf16 vs fp32: vit %.3e hidden(last) %.3e logits %.3e" % (rel(v16, v32), rel(h16[-1], h32[-1]), rel(l16, l32)))
    out["vit_fp32"], out["vit_bf16"] = v32.numpy(), bits(v16)
    for k in (4, 8, 12):
        out[f"hidden{k}_fp32"], out[f"hidden{k}_bf16"] = h32[k].numpy(), bits(h16[k])
    nt = 32                                                                    # logits of the text rows behind the video
    out["logits_tail_fp32"], out["logits_tail_bf16"] = l32[-nt:].numpy(), bits(l16[-nt:])
    for k, t in caps.items():
        out["pin_" + k] = bits(t)
    out["position_ids"] = pos.numpy()
    out["param_names"] = np.array(sorted(shapes))
    out["param_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes)])
    np.savez_compressed(os.path.join(OUT, "qwen_mid_bf16.npz"), **out)
    print("wrote", os.path.join(OUT, "qwen_mid_bf16.npz"), os.path.getsize(os.path.join(OUT, "qwen_mid_bf16.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
