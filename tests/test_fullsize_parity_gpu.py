"""GPU, BASELINE.json FULL sizes against the oracle (missing item 7): the HIP path and the fp32 restatement on the same bf16-rounded weights and
seeded inputs, at the 7B / SAM2-L dimensions the benchmarks run, forward and backward.  Each oracle leg takes a few seconds to a minute
on the box's host cores.  Tolerances are the stated ones: hidden states / logits rel-L2 <= 2e-2, gradients <= 3e-2.

  * one 7B decoder layer (+ final norm) at S = 2112, 28 Q / 4 KV heads x 128, mRoPE with real 3-axis video positions;
  * Qwen2.5-VL ViT at grid [8,32,32]: patch embed, one windowed + one full-attention block, merger (8192 patches -> 2048 x 3584);
  * attention backward at S = 2112 and S = 4160, 28 Q / 4 KV heads (the f32 dK/dV workspace route), vs autograd through the oracle's softmax attention;
  * one SAM2-L frame through Hiera-L (48 blocks) + FPN;
  * SAM2-L memory attention (4 layers) over the full bank: 7 x 4096 memory tokens + 64 pointer tokens = 28 736 keys.
"""
import numpy as np
import pytest
import torch

from oracle import kernels_ref as R
from oracle import qwen25vl as Q
from oracle import sam2 as S

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _threads():
    import os
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))


def _init(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif "norm" in n or "ln_q" in n:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return mod


def test_decoder_layer_7b_s2112(dev):
    from rga3.model import qwen_index as QI
    from rga3.model.qwen2_5_vl import Qwen2_5_VLConfig, TextModel

    _threads()
    c = Qwen2_5_VLConfig(num_hidden_layers=1)
    tm = _init(TextModel(c), 11)
    S_ = 2112
    ids = np.concatenate([np.arange(14), [c.vision_start_token_id], np.full(2048, c.video_token_id), [c.vision_end_token_id], np.arange(100, 148)])[None]
    pos_np, _ = QI.rope_index(ids, c.image_token_id, c.video_token_id, 2, 2, None, np.array([[8, 32, 32]]), np.array([1.0]), None, c.mrope_temporal_rule)
    x = (torch.randn(S_, c.hidden_size, generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16)
    P = {"model." + k: v.detach().to(torch.bfloat16).float() for k, v in tm.state_dict().items() if not k.startswith("embed_tokens")}
    tmd = tm.to(torch.bfloat16).to(dev).eval()
    pos3 = torch.from_numpy(pos_np.reshape(3, -1)).to(dev)
    cu = torch.tensor([0, S_], dtype=torch.int32, device=dev)
    with torch.no_grad():
        y, _ = tmd(x.to(dev), pos3, cu, S_)
        cfg = Q.QwenCfg(vision=Q.VisionCfg(depth=0, fullatt_block_indexes=()), text=Q.TextCfg(num_hidden_layers=1))
        ref = Q.llm_forward(P, x.float()[None], torch.from_numpy(pos_np), None, cfg)[0]
    assert rel(y, ref) < 2e-2


@pytest.mark.parametrize("frames", [8, 16])
def test_vit_blocks_7b_grid_8_32_32(dev, frames):
    """grid [8,32,32] = configs[1] / [2] (16 frames of 448 x 448); grid [16,32,32] = configs[4] (32 frames: 16 384 patches, 16 full-attention segments of 1024)."""
    from rga3.model.qwen2_5_vl import Qwen2_5_VLVisionConfig, VisionTransformer

    _threads()
    vc = Qwen2_5_VLVisionConfig(depth=2, fullatt_block_indexes=(1,))
    vt = _init(VisionTransformer(vc), 12)
    px = torch.randn(frames * 1024, 1176, generator=torch.Generator().manual_seed(4)).clamp_(-1.8, 2.2).to(torch.bfloat16)
    grid = np.array([[frames, 32, 32]])
    P = {"visual." + k: v.detach().to(torch.bfloat16).float() for k, v in vt.state_dict().items()}
    vtd = vt.to(torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        y = vtd(px.to(dev), grid)
        ref = Q.vit_forward(P, px.float(), grid, Q.QwenCfg(vision=Q.VisionCfg(depth=2, fullatt_block_indexes=(1,)), text=Q.TextCfg(num_hidden_layers=1)))
    assert tuple(y.shape) == (frames * 256, 3584)
    assert rel(y, ref) < 2e-2


def test_vit_blocks_7b_fp8_grid_16_32_32(dev):
    """configs[4] (fp8 LoRA fine-tune, 32 frames -> grid [16,32,32], 16 384 patches): the frozen vision tower's qkv / proj / gate|up / down contractions in e4m3
    (rga3.model.qwen_train.vision_block_forward_fp8; reference run: the frozen tower under run_torchrun.sh:28-40) on one windowed + one full-attention block at the 7B
    tower's dimensions, against the oracle's e4m3 restatement (oracle/fp8step.py with vision=True: same quantiser, same scales, products summed in fp32).  Bound: 1.25 x
    what e4m3 itself costs these two blocks (the oracle's e4m3 run against its fp32 run, printed), never less than the bf16 tolerance 2e-2; and the e4m3 route must
    really have run (its weight packs exist, its output differs from the bf16 route's)."""
    from rga3.model import qwen_train as QT
    from rga3.model.qwen2_5_vl import Qwen2_5_VLVisionConfig, VisionTransformer
    from oracle.fp8step import fp8_frozen_linears

    _threads()
    vc = Qwen2_5_VLVisionConfig(depth=2, fullatt_block_indexes=(1,))
    vt = _init(VisionTransformer(vc), 12)
    px = torch.randn(16 * 1024, 1176, generator=torch.Generator().manual_seed(4)).clamp_(-1.8, 2.2).to(torch.bfloat16)
    grid = np.array([[16, 32, 32]])
    P = {"visual." + k: v.detach().to(torch.bfloat16).float() for k, v in vt.state_dict().items()}
    vtd = vt.to(torch.bfloat16).to(dev).eval()
    ocfg = Q.QwenCfg(vision=Q.VisionCfg(depth=2, fullatt_block_indexes=(1,)), text=Q.TextCfg(num_hidden_layers=1))
    with torch.no_grad():
        y16 = vtd(px.to(dev), grid)
        QT.set_fp8_frozen_gemms(True)
        try:
            y8 = vtd(px.to(dev), grid)
        finally:
            QT.set_fp8_frozen_gemms(False)
        with fp8_frozen_linears(vision=True):
            r8 = Q.vit_forward(P, px.float(), grid, ocfg)
        r32 = Q.vit_forward(P, px.float(), grid, ocfg)
    assert any(k.startswith("fp8:") for k in vtd.blocks[0].mlp.__dict__.get("_wt_cache", {})), "the e4m3 weight packs were not built: the bf16 route ran"
    e8, yard, e16 = rel(y8, r8), rel(r8, r32), rel(y16, r32)
    print("VIT_FP8", {"fp8_vs_oracle_fp8": e8, "oracle_fp8_vs_oracle_fp32": yard, "bf16_vs_oracle_fp32": e16, "fp8_vs_bf16_product": rel(y8, y16)})
    assert tuple(y8.shape) == (4096, 3584) and e16 < 2e-2
    assert e8 < max(2e-2, 1.25 * yard), (e8, yard)
    assert rel(y8, y16) > 0.25 * yard                      # the two routes are different arithmetic


@pytest.mark.parametrize("S_", [2112, 4160])
def test_attention_backward_gqa_fullsize(dev, S_):
    """dq / dk / dv at the decoder's shapes (28 Q heads over 4 KV heads x 128: the per-query-head dK/dV workspace + fixed-order reduce) against autograd
    through the oracle's exact softmax attention."""
    from rga3.hip import ops

    _threads()
    Hq, Hk, D = 28, 4, 128
    g = torch.Generator().manual_seed(S_)
    q = (torch.randn(S_, Hq, D, generator=g) * 0.7).to(torch.bfloat16)
    k = (torch.randn(S_, Hk, D, generator=g) * 0.7).to(torch.bfloat16)
    v = (torch.randn(S_, Hk, D, generator=g) * 0.7).to(torch.bfloat16)
    do = (torch.randn(S_, Hq, D, generator=g) * 0.5).to(torch.bfloat16)
    cu = torch.tensor([0, S_], dtype=torch.int32)
    qd, kd, vd, dod, cud = (t.to(dev) for t in (q, k, v, do, cu))
    o, lse = ops.attn_varlen(qd, kd, vd, cud, cud, S_, D ** -0.5, causal=True, return_lse=True)
    dq, dk, dv = ops.attn_varlen_bwd(qd, kd, vd, o, dod, lse, cud, cud, S_, S_, D ** -0.5, True)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    ro, rlse = R.attn_varlen_ref(qf, kf, vf, cu, cu, D ** -0.5, True)
    ro.backward(do.float())
    assert rel(o, ro) < 1e-2 and rel(lse, rlse) < 1e-3
    assert rel(dq, qf.grad) < 2e-2 and rel(dk, kf.grad) < 2e-2 and rel(dv, vf.grad) < 2e-2


def _sam2_l(dev, seed):
    from rga3.model.sam2 import SAM2

    m = SAM2()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.02 if p.shape[-1] > 8 else 0.2))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    P = {k: v.detach().to(torch.bfloat16).float() for k, v in m.sam2_model.state_dict().items()}
    return m.to(torch.bfloat16).to(dev).eval(), P


def test_sam2_l_frame_hiera_fpn(dev):
    _threads()
    m, P = _sam2_l(dev, 21)
    from tests.blob_inputs import object_video
    img = object_video("fullsize_frame", 1, 1024, seed=1)[0].to(torch.bfloat16)
    with torch.no_grad():
        f = m.sam2_model.forward_image(img.to(dev))
        bo = S.image_encoder_forward(P, img.float(), S.Sam2Cfg())
    t2m = lambda t, H, W: t.float().cpu().view(1, H, W, -1).permute(0, 3, 1, 2)
    assert rel(t2m(f["feat"], 64, 64), bo["backbone_fpn"][2]) < 2e-2
    assert rel(t2m(f["feat_s1"], 128, 128), bo["backbone_fpn"][1]) < 2e-2
    assert rel(t2m(f["feat_s0"], 256, 256), bo["backbone_fpn"][0]) < 2e-2


def test_sam2_l_memory_attention_full_bank(dev):
    """4 layers, 4096 queries x (7 x 4096 + 64) keys of 64-d memory (k / v projected 64 -> 256 per layer), RoPE on the 28 672 spatial keys, the 64 pointer
    tokens excluded (reference sam2.py:1527-1533)."""
    _threads()
    m, P = _sam2_l(dev, 22)
    g = torch.Generator().manual_seed(5)
    nq, nk, nptr = 4096, 7 * 4096 + 64, 64
    curr = (torch.randn(nq, 256, generator=g)).to(torch.bfloat16)
    cpos = (torch.randn(nq, 256, generator=g)).to(torch.bfloat16)
    mem = (torch.randn(nk, 64, generator=g)).to(torch.bfloat16)
    mpos = (torch.randn(nk, 64, generator=g) * 0.5).to(torch.bfloat16)
    with torch.no_grad():
        y = m.sam2_model.memory_attention(curr.to(dev), cpos.to(dev), mem.to(dev), mpos.to(dev), nptr)
        ref = S.memory_attention(P, curr.float()[:, None], cpos.float()[:, None], mem.float()[:, None], mpos.float()[:, None], nptr, S.Sam2Cfg())[:, 0]
    assert rel(y, ref) < 2e-2


def test_sam2_l_concurrent_slot_graphs_bit_exact_at_bench_shape(dev):
    """ADVICE r5 (high) at the configs[3] shape (SAM2-L, 1024 x 1024: the memory cross-attention runs its split form with partial sums in scratch, the products their
    stream-K slabs): three object slots replaying already-captured frame graphs on three streams, clips 2 and 3 of the session cache, equal bit for bit to the same
    objects tracked one after the other without graphs."""
    from rga3.model.sam2 import MultiObjectSession, VideoSession

    m, _ = _sam2_l(dev, 23)
    n_obj, T = 3, 6
    g = torch.Generator().manual_seed(3)
    embs = [torch.randn(1, 1, 256, generator=g).to(torch.bfloat16).to(dev) for _ in range(n_obj)]

    def track(vid, feats, **kw):
        ms = MultiObjectSession(m.sam2_model, vid, n_obj, feats=feats)
        for o in range(n_obj):
            ms.add_language_embd(0, o, embs[o])
        return torch.cat([mk for _, mk in ms.propagate(**kw)], 0)

    with torch.no_grad():
        for clip in range(5):
            _shift = [torch.cuda.Stream() for _ in range(7 * clip % 32)]      # walk the 32-handle stream pool: eager streams meet the slots' capture streams
            vid = torch.randn(T, 3, 1024, 1024, generator=g).to(torch.bfloat16).to(dev)
            feats = VideoSession(m.sam2_model, vid)._ensure_feats()
            got = track(vid, feats, use_graph=True, concurrent=True)
            want = track(vid, feats, use_graph=False, concurrent=False)
            torch.cuda.synchronize()
            assert torch.equal(got, want), f"clip {clip}"
    assert not torch.equal(got[0], got[1])


def test_decoder_layer_7b_lora_r128_forward_backward_s2112(dev):
    """the TRAINING leg at full size.  One Qwen2.5-7B decoder layer (3584 wide, 28 Q / 4 KV heads x 128, SwiGLU 18 944) with LoRA r = 128 /
    alpha = 256 on q_proj and v_proj (reference train_joint.py:193-251, run_torchrun.sh:30-31), final RMSNorm, a trainable LM head (vocabulary cut to 8 192 rows
    so the fp32 oracle finishes in a minute) and embed_tokens, at S = 2112 with real 3-axis video positions: loss and the gradients dA, dB (q and v), d(lm_head),
    d(embed_tokens) -- every id occurs once, so the table's gradient rows ARE dX through RMSNorm / attention / SwiGLU backward -- against fp32 autograd through the
    oracle, at the flat 3e-2."""
    from rga3.model import qwen_index as QI
    from rga3.model.qwen2_5_vl import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    from rga3.model.qwen_train import add_lora

    _threads()
    V, S_ = 8192, 2112
    c = Qwen2_5_VLConfig(num_hidden_layers=1, vocab_size=V, vision_config={"depth": 1, "fullatt_block_indexes": (0,)})
    m = _init(Qwen2_5_VLForConditionalGeneration(c), 31)
    assert add_lora(m, r=128, alpha=256, dropout=0.0, exclude=("visual",)) == ["model.layers.0.self_attn.q_proj", "model.layers.0.self_attn.v_proj"]
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lora_" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    P = {k: v.detach().to(torch.bfloat16).float() for k, v in m.state_dict().items() if not k.startswith("visual.")}
    P = {k.replace(".base_layer.", "."): v for k, v in P.items()}
    P["lora_scaling"] = 2.0
    # positions of a 16-frame clip (grid [8,32,32] -> 2048 video tokens) inside 64 text tokens; the ids themselves are plain text ids so no vision tower runs
    ids_pos = np.concatenate([np.arange(14), [c.vision_start_token_id], np.full(2048, c.video_token_id), [c.vision_end_token_id], np.arange(100, 148)])[None]
    pos_np, _ = QI.rope_index(ids_pos, c.image_token_id, c.video_token_id, 2, 2, None, np.array([[8, 32, 32]]), np.array([1.0]), None, c.mrope_temporal_rule)
    ids = torch.randperm(V, generator=g)[:S_][None]
    labels = torch.full_like(ids, -100)
    labels[:, -64:] = ids[:, -64:]
    am = torch.ones_like(ids)
    md = m.to(torch.bfloat16).to(dev).train()
    train = [n for n, p in md.named_parameters() if ("lora_" in n) or n in ("lm_head.weight", "model.embed_tokens.weight")]
    for n, p in md.named_parameters():
        p.requires_grad_(n in train)
    out = md(input_ids=ids.to(dev), attention_mask=am.to(dev), labels=labels.to(dev), position_ids=torch.from_numpy(pos_np).to(dev))
    out.loss.backward()
    cfg = Q.QwenCfg(vision=Q.VisionCfg(depth=0, fullatt_block_indexes=()), text=Q.TextCfg(num_hidden_layers=1, vocab_size=V))
    okeys = [n.replace(".base_layer.", ".") for n in train]
    for k in okeys:
        P[k].requires_grad_(True)
    ref = Q.forward(P, cfg, ids, am, position_ids=torch.from_numpy(pos_np), labels=labels)
    ref["loss"].backward()
    assert abs(out.loss.item() - ref["loss"].item()) / ref["loss"].item() < 1e-2
    got = dict(md.named_parameters())
    errs = {n: rel(got[n].grad, P[k].grad) for n, k in zip(train, okeys)}
    assert len(errs) == 6 and all(e < 3e-2 for e in errs.values()), errs
    used = ids[0]
    assert rel(got["model.embed_tokens.weight"].grad[used.to(dev)], P["model.embed_tokens.weight"].grad[used]) < 3e-2


def _mask_decoder_case(dev, feature_scale=1.0, debug=None, firm_relu=False, pe_seed=7, emulate=False):
    """(shared with tools/decoder_fullsize_grad.py) VERDICT r2 item 6(a), SAM2 side: the trainable tail of the mask path at SAM2-L dimensions -- conv_s0 / conv_s1 on the 256^2 / 128^2 FPN levels, prompt tokens,
    two-way transformer over 4096 image tokens, 2 x ConvTranspose + LayerNorm2d + GELU to 256^2 x 32, hyper-network product, selected-mask bilinear 1024^2 -> label size,
    BCE + dice (reference sam2.py:1926-2210, qwen_2_5_vl_sam2.py:267-308) -- forward AND backward on 4 frames in one launch sequence (rga3_mask_product_bwd,
    bilinear_bwd_gather, layernorm_bwd_rows, pixel-shuffle backward at full size), against fp32 autograd through oracle/sam2.py, flat 3e-2 per tensor."""
    from rga3.hip import autograd as AG
    from rga3.model.sam2 import SAM2
    from oracle import unigr as U
    from tests.blob_inputs import masks_at, object_video

    _threads()
    B, h, w = 4, 64, 64
    m = SAM2()
    g = torch.Generator().manual_seed(41)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.shape[-1] > 8 else 0.2))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        m.sam2_model.sam_mask_decoder.iou_prediction_head.layers[2].bias.copy_(torch.tensor([0.0, -2.0, 2.0, -2.0]))   # a clear argmax: candidate 2 on every frame
        # the random-Fourier positional matrix is a BUFFER drawn from the global generator at construction: pin it, or every call is a different test point
        pe = m.sam2_model.sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix
        pe.copy_(torch.randn(pe.shape, generator=torch.Generator().manual_seed(pe_seed)))
        if firm_relu:
            # every ReLU of the token-side MLPs firmly on or off (|bias| >> |w x|): the bf16 forward and the fp32 oracle then agree on every unit's state, so the
            # comparison measures the backward kernels instead of which of a few hundred units sat within rounding noise of zero (each flipped unit of a
            # 256-wide single-row MLP moves that row's input gradient by several per cent)
            for n, p in m.named_parameters():
                if "sam_mask_decoder" in n and n.endswith(".bias") and (".mlp.layers.0." in n or "output_hypernetworks_mlps" in n and not n.endswith("layers.2.bias")):
                    p.copy_(torch.where(torch.rand(p.shape, generator=g) < 0.5, -1.0, 1.0) * (5.0 + torch.rand(p.shape, generator=g)))
    PS = {k: v.detach().to(torch.bfloat16).float() for k, v in m.sam2_model.state_dict().items()}
    sm = m.to(torch.bfloat16).to(dev).sam2_model
    names = [n for n, _ in sm.named_parameters() if n.startswith("sam_mask_decoder.")]
    for n, p in sm.named_parameters():
        p.requires_grad_(n in names)
    # smooth, object-like feature maps (a random field would make every candidate mask speckle): low-pass filtered noise at the three FPN resolutions
    def field(c, s, seed):
        z = torch.randn(B, c, s // 8, s // 8, generator=torch.Generator().manual_seed(seed))
        return (feature_scale * torch.nn.functional.interpolate(z, size=(s, s), mode="bicubic", align_corners=False)).to(torch.bfloat16)
    f2, f1, f0 = field(256, 64, 1), field(256, 128, 2), field(256, 256, 3)
    emb = (torch.randn(B, 1, 256, generator=g) * 0.5).to(torch.bfloat16)
    clip = object_video("fullsize_decoder", B, 1024, seed=2)
    gt = masks_at(clip[1], (480, 640)).float()
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    embd = emb.to(dev).requires_grad_(True)
    dec = sm.sam_mask_decoder
    feats = {"feat_s0": AG.linear(tok(f0).to(dev), dec.conv_s0.weight.reshape(dec.conv_s0.weight.shape[0], -1), dec.conv_s0.bias),
             "feat_s1": AG.linear(tok(f1).to(dev), dec.conv_s1.weight.reshape(dec.conv_s1.weight.shape[0], -1), dec.conv_s1.bias),
             "feat": tok(f2).to(dev), "hw": (h, w), "n": B, "pos": None}
    from rga3.hip import ops
    pix = ops.add_bcast(feats["feat"], sm.no_mem_embed.view(1, -1))
    if debug is not None:
        import rga3.model.sam2 as PS2
        PS2._DEBUG = debug.setdefault("product", {})
    o = sm.forward_sam_heads(pix, feats, embd)
    pred = AG.BilinearFn.apply(o["high_res_masks"][:, 0].contiguous(), (480, 640), None)
    bce, dice = AG.MaskLossFn.apply(pred, gt.to(dev))
    loss = 2.0 * bce / B + 0.5 * dice / B
    loss.backward()
    # ---- oracle
    cfg = S.Sam2Cfg()
    for n in names:
        PS[n].requires_grad_(True)
    embf = emb.float().requires_grad_(True)
    import torch.nn.functional as F
    high = [F.conv2d(f0.float(), PS["sam_mask_decoder.conv_s0.weight"], PS["sam_mask_decoder.conv_s0.bias"]),
            F.conv2d(f1.float(), PS["sam_mask_decoder.conv_s1.weight"], PS["sam_mask_decoder.conv_s1.bias"])]
    pixf = f2.float() + PS["no_mem_embed"].view(1, -1, 1, 1)
    ro = S.forward_sam_heads(PS, pixf, high, embf, cfg, True, None if debug is None else debug.setdefault("oracle", {}))
    rpred = F.interpolate(ro["high_res_masks"], size=(480, 640), mode="bilinear", align_corners=False)[:, 0]
    rloss = 2.0 * U.sigmoid_ce_loss(rpred, gt, B) + 0.5 * U.dice_loss(rpred, gt, B)
    rloss.backward()
    if debug is not None:
        import rga3.model.sam2 as PS2
        PS2._DEBUG = None
    got = dict(sm.named_parameters())
    # k_proj.bias: a constant added to every key shifts all scores of a query equally -- softmax is invariant, the exact gradient is 0 and both sides hold rounding noise
    errs = {n: rel(got[n].grad, PS[n].grad) for n in names if PS[n].grad is not None and PS[n].grad.norm() > 0 and not n.endswith("k_proj.bias")}
    errs["language_embd"] = rel(embd.grad, embf.grad)
    norms = {n: float(PS[n].grad.float().norm()) for n in errs if n != "language_embd"}
    norms["language_embd"] = float(embf.grad.norm())
    emu = None
    if emulate:     # what bf16 STORAGE of activations and gradients alone costs at this point (fp32 arithmetic inside every op): oracle vs oracle
        from oracle.bf16emu import bf16_storage
        PE = {k: v.detach().clone() for k, v in PS.items()}
        for n in names:
            PE[n].requires_grad_(True)
        embe = emb.float().requires_grad_(True)
        with bf16_storage():
            highe = [F.conv2d(f0.float(), PE["sam_mask_decoder.conv_s0.weight"], PE["sam_mask_decoder.conv_s0.bias"]),
                     F.conv2d(f1.float(), PE["sam_mask_decoder.conv_s1.weight"], PE["sam_mask_decoder.conv_s1.bias"])]
            roe = S.forward_sam_heads(PE, f2.float() + PE["no_mem_embed"].view(1, -1, 1, 1), highe, embe, cfg, True)
            rpe = F.interpolate(roe["high_res_masks"], size=(480, 640), mode="bilinear", align_corners=False)[:, 0]
            (2.0 * U.sigmoid_ce_loss(rpe, gt, B) + 0.5 * U.dice_loss(rpe, gt, B)).backward()
        emu = {n: rel(PE[n].grad, PS[n].grad) for n in errs if n != "language_embd"}
        emu["language_embd"] = rel(embe.grad, embf.grad)
    return dict(emu=emu, norms=norms, best=(o["best_iou_inds"].cpu(), ro["best_iou_inds"]), ious=(o["ious"], ro["ious"]), low=rel(o["low_res_masks"], ro["low_res_masks"]),
                loss=(loss.item(), rloss.item()), errs=errs)


def _check_forward(r):
    assert torch.equal(*r["best"]), r["ious"]
    assert r["low"] < 2e-2
    assert abs(r["loss"][0] - r["loss"][1]) / r["loss"][1] < 1e-2


@pytest.mark.parametrize("pe_seed", [7, 3])
def test_mask_decoder_sam2_l_forward_backward(dev, pe_seed):
    """Well-conditioned point: every ReLU of the token-side MLPs firmly on or off and a pinned positional matrix.  Every parameter gradient that
    carries more than 1e-5 of the total gradient norm is within the FLAT 3e-2 of fp32 autograd (measured: <= 0.9e-2 on 6 of 8 positional-matrix seeds,
    profiles/r03_decoder_grad_by_seed.log); the few below that share are analytically-near-zero products (image-to-token attention scores over 9 tokens) and are
    bounded in absolute size."""
    r = _mask_decoder_case(dev, firm_relu=True, pe_seed=pe_seed)     # two of the six well-conditioned positional-matrix seeds (not one pinned seed)
    _check_forward(r)
    errs, norms = r["errs"], r["norms"]
    tot = sum(v * v for v in norms.values()) ** 0.5
    sig = {n: e for n, e in errs.items() if norms[n] > 1e-5 * tot}
    bad = {n: round(e, 4) for n, e in sig.items() if e >= 3e-2}
    assert len(sig) > 80 and not bad, (bad, len(sig))
    for n, e in errs.items():     # negligible tensors: error small against the whole gradient
        if n not in sig:
            assert e * norms[n] < 1e-4 * tot, (n, e, norms[n] / tot)


def test_mask_decoder_sam2_l_backward_ill_conditioned_point(dev):
    """The same decoder with positional-matrix seed 6, one of the two seeds in eight where the token-path gradients of ANY bf16 pipeline sit 5 - 9 % from fp32 autograd:
    the fp32 oracle with nothing but bf16 STORAGE of activations and gradients (oracle/bf16emu.py; fp32 arithmetic inside every op) is itself 9 % off there, tensor by
    tensor in the same pattern as the HIP path.  Asserted: (a) that fact (so nobody mistakes the point for a kernel defect), (b) the HIP path stays within 1.5x the
    emulation's deviation + 1e-2 per significant tensor, under a hard ceiling of 0.15."""
    r = _mask_decoder_case(dev, firm_relu=True, pe_seed=6, emulate=True)
    _check_forward(r)
    errs, emu, norms = r["errs"], r["emu"], r["norms"]
    tot = sum(v * v for v in norms.values()) ** 0.5
    sig = [n for n in errs if norms[n] > 1e-5 * tot]
    assert max(emu[n] for n in sig) > 3e-2, "bf16 storage alone is within 3e-2 here: tighten this test to the flat bound"
    bad = {n: (round(errs[n], 4), round(emu[n], 4)) for n in sig if errs[n] > min(0.15, 1.2 * emu[n] + 1e-2)}     # (1.5 x until round 4; measured ratio <= 1.01 on the large tensors)
    assert not bad, bad


def test_mask_decoder_sam2_l_backward_random_point(dev):
    """A generic random point (MLP biases near zero: ReLU units within bf16 rounding of zero are on in one arithmetic and off in the other, on top of the above).  Forward
    parity at the stated tolerance; every gradient tensor with more than 1e-3 of the gradient norm within 1.5x what bf16 storage alone costs at this point + 1e-2, under a
    hard ceiling of 0.25; whole-gradient relative error < 0.15."""
    r = _mask_decoder_case(dev, firm_relu=False, pe_seed=7, emulate=True)
    _check_forward(r)
    errs, emu, norms = r["errs"], r["emu"], r["norms"]
    tot = sum(v * v for v in norms.values()) ** 0.5
    bad = {n: (round(errs[n], 4), round(emu[n], 4)) for n in errs if norms[n] > 1e-3 * tot and errs[n] > min(0.25, 1.5 * emu[n] + 1e-2)}
    assert not bad, bad
    ratio = (sum((errs[n] * norms[n]) ** 2 for n in errs) ** 0.5) / tot
    assert ratio < 0.15, ratio
