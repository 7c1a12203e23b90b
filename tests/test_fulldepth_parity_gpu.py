"""GPU, BASELINE.json configs[1] at FULL DEPTH against the oracle: the whole Qwen2.5-VL-7B forward -- 32 ViT blocks + merger, embedding scatter,
28 decoder layers with mRoPE, final norm, LM head over the 152 064-row vocabulary -- on 16 frames of 448 x 448 (grid [8,32,32], S = 2112) in ONE call of the product
model (reference model/qwen_2_5_vl_sam2.py:182-200 -> HF modeling_qwen2_5_vl.py:1185-1253, 1367-1402), against oracle/qwen25vl.py in fp32 on the host cores on the
same bf16 weights.  north_star's acceptance: "bit-exact token indices, fp logits within stated tolerance" (logits rel-L2 <= 2e-2 -- met by the
32-block vision tower and by every single layer; at the end of all 60 blocks the bound is max(2e-2, 1.25 x what bf16 STORAGE alone costs the fp32 restatement),
see the yardstick below).

The weights are random (no checkpoint offline) but CONDITIONED: residual-branch output projections are scaled by 1 / sqrt(2 L) (the GPT-2 initialisation), so the
60-block stack does not amplify perturbations the way an N(0, 0.02) stack does (bench.py's random-init forward drifts 9 % between two tilings of the SAME product
path -- that measures chaos, not arithmetic).  The oracle never holds more than one weight matrix in fp32: its parameter dictionary fetches each tensor from the
device model on demand (the host memory of the GPU box is not ours to fill with 33 GB).

Token indices: greedy argmax must equal the oracle's on every row whose top-1 margin exceeds 8 x the measured RMS logit error (rows the comparison can decide);
the agreement over ALL rows is reported.

Second leg, configs[4] (fp8 LoRA step, 32 frames -> grid [16,32,32], S = 4160) on one 7B decoder layer: forward, loss and gradients with the frozen contractions
in e4m3 against the oracle's e4m3 restatement (oracle/fp8step.py) at M = 4160; the e4m3 products at that M are pinned in tests/test_kernels_gpu.py, the ViT at
16 384 patches in tests/test_fullsize_parity_gpu.py."""
import json
import os
import time

import numpy as np
import pytest
import torch

from oracle import qwen25vl as Q

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _threads():
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))


class DeviceParams(dict):
    """The oracle's parameter dictionary over the DEVICE model's bf16 tensors: a tensor is copied to the host and widened to fp32 when the oracle asks for it, and
    freed when the oracle drops it -- same rounded weights on both sides, one matrix resident at a time."""

    def __init__(self, sd):
        super().__init__()
        self._sd = sd

    def __getitem__(self, k):
        return self._sd[k].detach().float().cpu()

    def get(self, k, default=None):
        return self[k] if k in self._sd else default

    def __contains__(self, k):
        return k in self._sd


def _conditioned_init(model, seed):
    """N(0, 0.02) matrices, embedding rows N(0, 1) (a unit-scale residual stream from the first layer on), norm weights 1 + 0.1 N(0, 1), and every residual-branch
    output projection (attention out, MLP down; ViT and decoder) divided by sqrt(2 L) of its tower."""
    c = model.config
    Lv, Lt = c.vision_config.depth, c.num_hidden_layers
    g = torch.Generator(device=model.device).manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n == "model.embed_tokens.weight":
                p.normal_(0.0, 1.0, generator=g)
            elif p.dim() >= 2:
                p.normal_(0.0, 0.02, generator=g)
                if n.endswith(("attn.proj.weight", "mlp.down_proj.weight")) and n.startswith("visual.blocks."):
                    p.mul_((2 * Lv) ** -0.5)
                elif n.endswith(("self_attn.o_proj.weight", "mlp.down_proj.weight")) and n.startswith("model.layers."):
                    p.mul_((2 * Lt) ** -0.5)
            elif "norm" in n or "ln_q" in n:
                p.normal_(0.0, 0.1, generator=g).add_(1.0)
            else:
                p.normal_(0.0, 0.02, generator=g)


def _record(name, rec):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, name), "w") as f:
            json.dump(rec, f, indent=1)


def test_full_depth_7b_forward_16_frames(dev):
    from rga3.model.qwen2_5_vl import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration

    _threads()
    cfg = Qwen2_5_VLConfig()                       # the public Qwen2.5-VL-7B dimensions are the defaults
    assert cfg.num_hidden_layers == 28 and cfg.vision_config.depth == 32 and cfg.vocab_size == 152064
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            model = Qwen2_5_VLForConditionalGeneration(cfg)
    finally:
        torch.set_default_dtype(old)
    _conditioned_init(model, 71)
    model.eval()
    g = torch.Generator().manual_seed(72)
    grid = np.array([[8, 32, 32]])
    px = torch.randn(8192, 1176, generator=g).clamp_(-1.8, 2.2).to(torch.bfloat16)
    text = torch.randint(0, 151643, (64,), generator=g)
    ids = torch.cat([text[:14], torch.tensor([cfg.vision_start_token_id]), torch.full((2048,), cfg.video_token_id), torch.tensor([cfg.vision_end_token_id]), text[16:]])[None]
    assert ids.shape[1] == 2112
    am = torch.ones_like(ids)
    with torch.no_grad():
        out = model(input_ids=ids.to(dev), attention_mask=am.to(dev), pixel_values_videos=px.to(dev), video_grid_thw=torch.from_numpy(grid),
                    second_per_grid_ts=torch.tensor([1.0]), output_hidden_states=True)
        vit_dev = model.visual(px.to(dev), grid)
    logits = out.logits[0].float().cpu()
    hidden = out.hidden_states[-1][0].float().cpu()
    assert tuple(logits.shape) == (2112, 152064)

    # ---- the oracle, piece by piece as Q.forward runs them (HF:1185-1253, 1367-1402), so that the towers' errors are reported separately
    P = DeviceParams(model.state_dict())
    ocfg = Q.QwenCfg()
    t0 = time.time()
    with torch.no_grad():
        e = Q.vit_forward(P, px.float(), grid, ocfg)
        t_vit = time.time() - t0
        x = P["model.embed_tokens.weight"][ids]
        x[ids == ocfg.video_token_id] = e
        pos, _ = Q.rope_index(ids.numpy(), ocfg, None, grid, np.array([1.0]), am.numpy(), "hf449")
        hs_ref = []
        hid_ref = Q.llm_forward(P, x, torch.from_numpy(pos), am, ocfg, hidden_states=hs_ref)[0]
        hs_ref = [h[0] for h in hs_ref]
        log_ref = hid_ref @ P["lm_head.weight"].t()
    t_all = time.time() - t0
    # ---- the yardstick: the SAME restatement with every module output rounded to bf16 storage, the way the reference runs the model (app.py:53-58, train_joint.py:
    #      165-179; oracle.qwen25vl.storage).  What bf16 storage alone does to 60 residual blocks is not a property of any kernel: ~10 roundings of 2^-9 / sqrt(3) per
    #      block accumulate as a random walk to ~sqrt(60) x 3.5e-3.  PINNED since round 6: tests/golden/qwen_mid_bf16.npz holds transformers' own bf16 (and fp32)
    #      run at a mid size; tests/test_oracle_qwen.py checks that this mode reproduces it block by block (2 - 5e-4, against 2.3e-3 for the fp32 mode) and that the
    #      yardstick's size equals what the bf16 run costs transformers itself to 2 % at five depths.  Still used as a yardstick only, never as the oracle.
    with torch.no_grad(), Q.storage(torch.bfloat16):
        e16 = Q.vit_forward(P, px.float(), grid, ocfg)
        x16 = P["model.embed_tokens.weight"][ids]
        x16[ids == ocfg.video_token_id] = e16
        hs16 = []
        hid16 = Q.llm_forward(P, x16, torch.from_numpy(pos), am, ocfg, hidden_states=hs16)[0]
        hs16 = [h[0] for h in hs16]
        log16 = (hid16 @ P["lm_head.weight"].t()).to(torch.bfloat16).float()
    y_vit, y_hid, y_log = rel(e16, e), rel(hid16, hid_ref), rel(log16, log_ref)
    # ---- growth along the depth: the residual stream after every 4th decoder layer, product vs fp32 oracle (e_k) next to the storage yardstick
    #      (y_k).  Rounding noise random-walks (the yardstick's own growth is pinned as such at mid size, tests/test_oracle_qwen.py) while a kernel with a SYSTEMATIC
    #      bias adds it coherently, e_k ~ k.  Two checks: (1) at every depth the product stays within 1.25 x the yardstick of THAT depth (not only at the end);
    #      (2) the ratio e_k / y_k beyond layer 8 never exceeds 1.15 x its value over the first 8 layers: against a sqrt(k) yardstick a linear-in-k error of equal size
    #      at k = 8 would stand at sqrt(24 / 8) = 1.7 x by layer 24.
    depths = [k for k in range(0, ocfg.text.num_hidden_layers + 1, 4)]
    dev_hs = [out.hidden_states[k][0].float().cpu() for k in depths]          # hidden_states[k]: the stream after k layers (the last entry is post-norm: taken from `hidden`)
    e_k = [rel(dev_hs[i], hs_ref[k]) if k < ocfg.text.num_hidden_layers else None for i, k in enumerate(depths)]
    y_k = [rel(hs16[k], hs_ref[k]) if k < ocfg.text.num_hidden_layers else None for k in depths]

    e_vit, e_hid, e_log = rel(vit_dev, e), rel(hidden, hid_ref), rel(logits, log_ref)
    sigma = float((logits - log_ref).pow(2).mean().sqrt())           # RMS logit error: "the measured error"
    top2 = log_ref.topk(2, dim=1)
    margin = top2.values[:, 0] - top2.values[:, 1]
    am_ref, am_dev = top2.indices[:, 0], logits.argmax(1)
    decisive = margin > 8.0 * sigma
    agree_all = float((am_ref == am_dev).float().mean())
    agree_dec = float((am_ref[decisive] == am_dev[decisive]).float().mean()) if bool(decisive.any()) else 1.0
    rec = {"vit_rel_l2": e_vit, "hidden_rel_l2": e_hid, "logits_rel_l2": e_log, "bf16_storage_yardstick": {"vit": y_vit, "hidden": y_hid, "logits": y_log},
           "vs_bf16_storage_oracle": {"vit": rel(vit_dev, e16), "hidden": rel(hidden, hid16), "logits": rel(logits, log16)},
           "logit_rms": float(log_ref.pow(2).mean().sqrt()), "logit_rms_err": sigma,
           "rows": 2112, "rows_decisive": int(decisive.sum()), "argmax_agree_decisive": agree_dec, "argmax_agree_all_rows": agree_all,
           "depth_profile": {"layers": depths, "product_vs_fp32": e_k, "storage_yardstick": y_k},
           "median_top1_margin": float(margin.median()), "oracle_seconds": round(t_all, 1), "oracle_vit_seconds": round(t_vit, 1), "oracle_threads": torch.get_num_threads()}
    print("FULL_DEPTH_7B", json.dumps(rec))
    _record("fulldepth_parity_7b.json", rec)
    # the stated tolerance (rel-L2 <= 2e-2) holds per tower entry (the 32-block ViT here; one decoder layer / two ViT blocks in
    # test_fullsize_parity_gpu.py); at the END of 60 blocks the bound is what bf16 storage itself costs the fp32 restatement, with 25 % headroom
    assert e_vit < max(2e-2, 1.25 * y_vit), rec
    assert e_hid < max(2e-2, 1.25 * y_hid) and e_log < max(2e-2, 1.25 * y_log), rec
    # fixed ceilings beside the relative bound: the yardstick is pinned to transformers' bf16 run at mid size (tests/test_oracle_qwen.py), and here it must
    # stay inside the range 60 blocks of bf16 storage give (measured 1.46e-2 / 2.78e-2 / 2.77e-2) -- a change of the oracle cannot silently move the gate
    assert 1.1e-2 < y_vit < 1.9e-2 and 2.2e-2 < y_hid < 3.4e-2 and 2.2e-2 < y_log < 3.4e-2, rec
    assert e_vit < 2e-2 and e_hid < 3.4e-2 and e_log < 3.4e-2, rec
    assert rec["vs_bf16_storage_oracle"]["logits"] < 3.4e-2 and rec["vs_bf16_storage_oracle"]["hidden"] < 3.4e-2, rec
    # growth along the depth
    pts = [(k, ek, yk) for k, ek, yk in zip(depths, e_k, y_k) if ek is not None and k > 0]
    for k, ek, yk in pts:
        assert ek < max(2e-2, 1.25 * yk), (k, ek, yk, rec["depth_profile"])
    ratio = {k: ek / yk for k, ek, yk in pts}
    early = max(ratio[4], ratio[8])
    for k, ek, yk in pts:
        if k > 8:      # ... and relative to the yardstick's own random walk the product's error must not GROW with depth (a coherent bias does: its ratio climbs like sqrt(k))
            assert ratio[k] <= 1.15 * early, (k, ratio, rec["depth_profile"])
    assert int(decisive.sum()) >= 200, rec                            # the token-index check must not be vacuous
    assert agree_dec == 1.0, rec                                      # bit-exact token indices wherever the comparison can decide
    assert agree_all >= 0.9, rec


def test_decoder_layer_7b_fp8_frozen_lora_r128_s4160(dev):
    """configs[4] at its own size: one 7B decoder layer, LoRA r = 128 / alpha = 256 on q and v, frozen q|k|v / o / gate|up / down contractions in e4m3 (reference
    run_torchrun.sh:28-40 fp8 fine-tune; PEFT LoRA under train_joint.py:193-251), S = 4160 = 4096 video tokens of grid [16,32,32] + 64 text tokens: loss and the six
    trainable gradients against fp32 autograd through the oracle's e4m3 restatement (oracle/fp8step.py: same quantiser, same scales, products summed in fp32).
    Tolerance: the loss at 1e-2; the gradients against what e4m3 itself costs this layer -- an activation that differs by one bf16 ulp between the two sides lands
    on the NEIGHBOURING e4m3 code (12.5 % apart) in ~3 % of its elements, ~2 % noise per contraction output and eight contractions deep (measured 9 - 10 %), so the
    bound is 1.25 x the distance between the oracle's own e4m3 step and its plain fp32 step on the same weights (both printed), never less than the bf16 layer's 3e-2.
    The e4m3 kernels themselves are pinned exactly at this M in tests/test_kernels_gpu.py::test_fp8_quant_and_gemm (same codes in, fp32 sums out)."""
    from rga3.model import qwen_index as QI
    from rga3.model import qwen_train as QT
    from rga3.model.qwen2_5_vl import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    from oracle.fp8step import fp8_frozen_linears, fp8_frozen_linears_on_codes

    _threads()
    V, S_ = 8192, 4160
    c = Qwen2_5_VLConfig(num_hidden_layers=1, vocab_size=V, vision_config={"depth": 1, "fullatt_block_indexes": (0,)})
    m = Qwen2_5_VLForConditionalGeneration(c)
    g = torch.Generator().manual_seed(81)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif "norm" in n or "ln_q" in n:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    assert QT.add_lora(m, r=128, alpha=256, dropout=0.0, exclude=("visual",)) == ["model.layers.0.self_attn.q_proj", "model.layers.0.self_attn.v_proj"]
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lora_" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    P = {k.replace(".base_layer.", "."): v.detach().to(torch.bfloat16).float() for k, v in m.state_dict().items() if not k.startswith("visual.")}
    P["lora_scaling"] = 2.0
    ids_pos = np.concatenate([np.arange(14), [c.vision_start_token_id], np.full(4096, c.video_token_id), [c.vision_end_token_id], np.arange(100, 148)])[None]
    pos_np, _ = QI.rope_index(ids_pos, c.image_token_id, c.video_token_id, 2, 2, None, np.array([[16, 32, 32]]), np.array([1.0]), None, c.mrope_temporal_rule)
    ids = torch.randperm(V, generator=g)[:S_][None]
    labels = torch.full_like(ids, -100)
    labels[:, -64:] = ids[:, -64:]
    am = torch.ones_like(ids)
    md = m.to(torch.bfloat16).to(dev).train()
    train = [n for n, p in md.named_parameters() if ("lora_" in n) or n in ("lm_head.weight", "model.embed_tokens.weight")]
    for n, p in md.named_parameters():
        p.requires_grad_(n in train)
    QT.set_fp8_frozen_gemms(True)
    codes = {}
    QT._fp8_tap[0] = lambda key, q_, s_: codes.__setitem__(key, (q_.clone(), s_.clone()))     # the e4m3 operand of every frozen contraction, as the build quantised it
    try:
        out = md(input_ids=ids.to(dev), attention_mask=am.to(dev), labels=labels.to(dev), position_ids=torch.from_numpy(pos_np).to(dev))
        out.loss.backward()
    finally:
        QT.set_fp8_frozen_gemms(False)
        QT._fp8_tap[0] = None
    assert set(codes) == {"wqkv", "wo", "wgu", "wd", "wqkv_t", "wo_t", "wgu_t", "wd_t"}, sorted(codes)
    assert any(k.startswith("fp8:") for k in md.model.layers[0].mlp.__dict__.get("_wt_cache", {})), "the e4m3 weight packs were not built: the bf16 route ran"
    cfg = Q.QwenCfg(vision=Q.VisionCfg(depth=0, fullatt_block_indexes=()), text=Q.TextCfg(num_hidden_layers=1, vocab_size=V))
    okeys = [n.replace(".base_layer.", ".") for n in train]
    for k in okeys:
        P[k].requires_grad_(True)
    with fp8_frozen_linears():
        ref = Q.forward(P, cfg, ids, am, position_ids=torch.from_numpy(pos_np), labels=labels)
        ref["loss"].backward()
    g8 = {k: P[k].grad.clone() for k in okeys}
    for k in okeys:
        P[k].grad = None
    ref32 = Q.forward(P, cfg, ids, am, position_ids=torch.from_numpy(pos_np), labels=labels)      # the same step without e4m3: what the quantisation costs
    ref32["loss"].backward()
    got = dict(md.named_parameters())
    errs = {n: rel(got[n].grad, g8[k]) for n, k in zip(train, okeys)}
    yard = {n: rel(g8[k], P[k].grad) for n, k in zip(train, okeys)}
    # ---- the same step once more on the build's OWN operand codes: no code flips between the two sides, what is left is kernel arithmetic
    for k in okeys:
        P[k].grad = None
    with fp8_frozen_linears_on_codes(codes):
        refc = Q.forward(P, cfg, ids, am, position_ids=torch.from_numpy(pos_np), labels=labels)
        refc["loss"].backward()
    errs_c = {n: rel(got[n].grad, P[k].grad) for n, k in zip(train, okeys)}
    rec = {"loss": out.loss.item(), "oracle_fp8_loss": ref["loss"].item(), "oracle_fp32_loss": ref32["loss"].item(), "grad_rel_l2": errs,
           "oracle_fp8_vs_oracle_fp32_grad_rel_l2": yard, "oracle_on_build_codes_loss": refc["loss"].item(), "grad_rel_l2_on_build_codes": errs_c}
    print("FP8_LAYER_S4160", json.dumps(rec))
    _record("fp8_layer_s4160_parity.json", rec)
    assert abs(out.loss.item() - ref["loss"].item()) / ref["loss"].item() < 1e-2, rec
    assert len(errs) == 6 and all(errs[n] < max(3e-2, 1.25 * yard[n]) for n in errs), rec
    # on shared codes the bound is the bf16 layer's (tests/test_fullsize_parity_gpu.py::test_decoder_layer_7b_lora_r128_forward_backward_s2112): a 5 % error in any
    # backward kernel of the e4m3 step now fails, where the own-codes comparison above could not see it under ~10 % of code-flip noise
    assert abs(out.loss.item() - refc["loss"].item()) / refc["loss"].item() < 5e-3, rec
    assert all(errs_c[n] < 3e-2 for n in errs_c), rec
