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


def test_vit_blocks_7b_grid_8_32_32(dev):
    from rga3.model.qwen2_5_vl import Qwen2_5_VLVisionConfig, VisionTransformer

    _threads()
    vc = Qwen2_5_VLVisionConfig(depth=2, fullatt_block_indexes=(1,))
    vt = _init(VisionTransformer(vc), 12)
    px = torch.randn(8192, 1176, generator=torch.Generator().manual_seed(4)).clamp_(-1.8, 2.2).to(torch.bfloat16)
    grid = np.array([[8, 32, 32]])
    P = {"visual." + k: v.detach().to(torch.bfloat16).float() for k, v in vt.state_dict().items()}
    vtd = vt.to(torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        y = vtd(px.to(dev), grid)
        ref = Q.vit_forward(P, px.float(), grid, Q.QwenCfg(vision=Q.VisionCfg(depth=2, fullatt_block_indexes=(1,)), text=Q.TextCfg(num_hidden_layers=1)))
    assert tuple(y.shape) == (2048, 3584)
    assert rel(y, ref) < 2e-2


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
