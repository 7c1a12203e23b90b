"""ORACLE — test infrastructure only (never imported by the product path).

fp32 PyTorch-CPU restatement of the SAM2 half of RGA3: reference model/sam2.py (vendored, 4174 lines).  Functional
style over a flat state dict whose keys are the reference's parameter names, so the same
deterministic weights can be poured into the reference's own classes (tests/golden/make_sam2_fixtures.py, build
container only) and into this restatement.  Pinned by tests/test_oracle_sam2.py against golden vectors produced by
importing /root/reference/model/sam2.py itself (G1-G3 of SURVEY.md 8(c)).

Every function cites the reference lines it follows ("S:n" = /root/reference/model/sam2.py line n).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

NO_OBJ_SCORE = -1024.0


@dataclass
class Sam2Cfg:
    """Defaults = SAM2-L as hard-coded in S:87-325."""
    image_size: int = 1024
    embed_dim: int = 144
    num_heads: int = 2
    stages: Sequence[int] = (2, 6, 36, 4)
    global_att_blocks: Sequence[int] = (23, 33, 43)
    window_spec: Sequence[int] = (8, 4, 16, 8)
    pos_bkg: Sequence[int] = (7, 7)
    q_pool: int = 3
    d_model: int = 256
    fpn_top_down_levels: Sequence[int] = (2, 3)
    scalp: int = 1
    backbone_stride: int = 16
    mem_dim: int = 64
    num_maskmem: int = 7
    memattn_layers: int = 4
    max_obj_ptrs_in_encoder: int = 16
    dec_depth: int = 2
    dec_heads: int = 8
    sigmoid_scale_for_mem_enc: float = 20.0
    sigmoid_bias_for_mem_enc: float = -10.0
    fuser_layers: int = 2

    @property
    def depth(self):
        return sum(self.stages)

    @property
    def stage_ends(self):
        return [sum(self.stages[:i]) - 1 for i in range(1, len(self.stages) + 1)]

    @property
    def q_pool_blocks(self):
        return [x + 1 for x in self.stage_ends[:-1]][: self.q_pool]

    def block_table(self):
        """Per-block (dim, dim_out, heads, window, pool) following the construction loop S:1182-1210
        (window lags the stage change by one block; heads/dim double at stage change)."""
        rows = []
        dim, heads, cur_stage = self.embed_dim, self.num_heads, 1
        for i in range(self.depth):
            dim_out = dim
            window = self.window_spec[cur_stage - 1]
            if i in self.global_att_blocks:
                window = 0
            if i - 1 in self.stage_ends:
                dim_out = dim * 2
                heads = heads * 2
                cur_stage += 1
            rows.append(dict(dim=dim, dim_out=dim_out, heads=heads, window=window, pool=i in self.q_pool_blocks))
            dim = dim_out
        return rows

    @property
    def channel_list(self):
        t = self.block_table()
        return [t[i]["dim_out"] for i in self.stage_ends[::-1]]


# ------------------------------------------------------------------------------------------------ small pieces
def lin(x, P, name):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def layer_norm(x, P, name, eps):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def layer_norm_2d(x, P, name, eps=1e-6):
    """S:2334-2346 channel-dim LayerNorm on NCHW."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return P[name + ".weight"][:, None, None] * x + P[name + ".bias"][:, None, None]


def mlp(x, P, name, n_layers, act=F.relu, sigmoid_output=False):
    """S:2305-2329"""
    for i in range(n_layers):
        x = lin(x, P, f"{name}.layers.{i}")
        if i < n_layers - 1:
            x = act(x)
    return torch.sigmoid(x) if sigmoid_output else x


def sdpa(q, k, v):
    """softmax(q k^T / sqrt(d)) v on [B, H, N, d] (F.scaled_dot_product_attention default scale)."""
    s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    return torch.softmax(s, dim=-1) @ v


def position_embedding_sine(num_pos_feats: int, h: int, w: int, temperature=10000.0):
    """S:1781-1814 (normalize=True, scale=2*pi). Returns [num_pos_feats, h, w]."""
    npf = num_pos_feats // 2
    y = torch.arange(1, h + 1, dtype=torch.float32).view(-1, 1).repeat(1, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, -1).repeat(h, 1)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)


def position_embedding_random(G: torch.Tensor, h: int, w: int):
    """S:1832-1856 dense PE from the saved gaussian matrix G [2, F]. Returns [2F, h, w]."""
    grid = torch.ones((h, w), dtype=torch.float32)
    y = (grid.cumsum(0) - 0.5) / h
    x = (grid.cumsum(1) - 0.5) / w
    c = 2 * torch.stack([x, y], dim=-1) - 1
    c = 2 * math.pi * (c @ G)
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).permute(2, 0, 1)


def get_1d_sine_pe(pos_inds, dim, temperature=10000.0):
    """S:2257-2267"""
    pe_dim = dim // 2
    dim_t = torch.arange(pe_dim, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / pe_dim)
    pe = pos_inds.unsqueeze(-1) / dim_t
    return torch.cat([pe.sin(), pe.cos()], dim=-1)


def compute_axial_cis(dim: int, end_x: int, end_y: int, theta=10000.0):
    """S:1874-1890 -> (cos, sin) each [end_x*end_y, dim/2] (real form of the complex table)."""
    fr = 1.0 / (theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
    t = torch.arange(end_x * end_y, dtype=torch.float32)
    tx, ty = (t % end_x), torch.div(t, end_x, rounding_mode="floor")
    ang = torch.cat([torch.outer(tx, fr), torch.outer(ty, fr)], dim=-1)
    return ang.cos(), ang.sin()


def apply_rotary_enc(xq, xk, cos, sin, repeat_freqs_k=False):
    """S:1901-1923: complex multiply on consecutive (even, odd) pairs, fp32. x [B, H, N, d]; cos/sin [Nq, d/2]."""
    def rot(x, c, s):
        xr, xi = x[..., 0::2], x[..., 1::2]
        return torch.stack([xr * c - xi * s, xr * s + xi * c], dim=-1).flatten(-2)
    q = rot(xq.float(), cos, sin)
    if xk.shape[-2] == 0:
        return q, xk
    if repeat_freqs_k:
        r = xk.shape[-2] // xq.shape[-2]
        cos, sin = cos.repeat(r, 1), sin.repeat(r, 1)
    return q, rot(xk.float(), cos, sin)


def select_closest_cond_frames(frame_idx, cond, max_num):
    """S:2212-2254"""
    if max_num == -1 or len(cond) <= max_num:
        return cond, {}
    sel = {}
    before = max((t for t in cond if t < frame_idx), default=None)
    if before is not None:
        sel[before] = cond[before]
    after = min((t for t in cond if t >= frame_idx), default=None)
    if after is not None:
        sel[after] = cond[after]
    rest = sorted((t for t in cond if t not in sel), key=lambda x: abs(x - frame_idx))[: max_num - len(sel)]
    sel.update((t, cond[t]) for t in rest)
    return sel, {t: v for t, v in cond.items() if t not in sel}


# ------------------------------------------------------------------------------------------------ Hiera + FPN
def window_partition(x, ws):
    """S:891-913"""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    """S:916-937"""
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def max_pool_2x2(x):
    """S:972-983 do_pool with MaxPool2d(2,2) on [B,H,W,C]."""
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def hiera_block(x, P, pre, row):
    """S:1085-1117 MultiScaleBlock + S:1007-1032 MultiScaleAttention."""
    heads, ws = row["heads"], row["window"]
    shortcut = x
    x = layer_norm(x, P, pre + "norm1", 1e-6)
    if row["dim"] != row["dim_out"]:
        shortcut = lin(x, P, pre + "proj")
        if row["pool"]:
            shortcut = max_pool_2x2(shortcut)
    H, W = x.shape[1], x.shape[2]
    pad_hw = (H, W)
    if ws > 0:
        x, pad_hw = window_partition(x, ws)
    B, h, w, _ = x.shape
    qkv = lin(x, P, pre + "attn.qkv").reshape(B, h * w, 3, heads, -1)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    if row["pool"]:
        q = max_pool_2x2(q.reshape(B, h, w, -1))
        h, w = q.shape[1:3]
        q = q.reshape(B, h * w, heads, -1)
    o = sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(B, h, w, -1)
    x = lin(o, P, pre + "attn.proj")
    if row["pool"]:
        ws = ws // 2
        H, W = shortcut.shape[1:3]
        pad_hw = (H + (ws - H % ws) % ws, W + (ws - W % ws) % ws) if ws > 0 else (H, W)
    if row["window"] > 0:
        x = window_unpartition(x, ws, pad_hw, (H, W))
    x = shortcut + x
    h2 = layer_norm(x, P, pre + "norm2", 1e-6)
    h2 = lin(F.gelu(lin(h2, P, pre + "mlp.layers.0")), P, pre + "mlp.layers.1")
    return x + h2


def hiera_forward(P, img, cfg: Sam2Cfg, pre="image_encoder.trunk."):
    """S:1228-1244 + S:1218-1226 (bicubic background pos-embed + tiled window embed)."""
    x = F.conv2d(img, P[pre + "patch_embed.proj.weight"], P[pre + "patch_embed.proj.bias"], stride=4, padding=3).permute(0, 2, 3, 1)
    h, w = x.shape[1:3]
    pe = F.interpolate(P[pre + "pos_embed"], size=(h, w), mode="bicubic")
    we = P[pre + "pos_embed_window"]
    pe = pe + we.tile([a // b for a, b in zip(pe.shape, we.shape)])
    x = x + pe.permute(0, 2, 3, 1)
    outs = []
    for i, row in enumerate(cfg.block_table()):
        x = hiera_block(x, P, f"{pre}blocks.{i}.", row)
        if i in cfg.stage_ends:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


def image_encoder_forward(P, img, cfg: Sam2Cfg):
    """S:785-798 ImageEncoder + S:857-889 FpnNeck + S:2790-2802 forward_image (conv_s0/conv_s1)."""
    xs = hiera_forward(P, img, cfg)
    n = len(xs) - 1
    out, pos = [None] * len(xs), [None] * len(xs)
    prev = None
    for i in range(n, -1, -1):
        lat = F.conv2d(xs[i], P[f"image_encoder.neck.convs.{n - i}.conv.weight"], P[f"image_encoder.neck.convs.{n - i}.conv.bias"])
        if i in cfg.fpn_top_down_levels and prev is not None:
            prev = lat + F.interpolate(prev.float(), scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
        pos[i] = position_embedding_sine(cfg.d_model, prev.shape[-2], prev.shape[-1])[None].repeat(prev.shape[0], 1, 1, 1)
    if cfg.scalp > 0:
        out, pos = out[: -cfg.scalp], pos[: -cfg.scalp]
    out[0] = F.conv2d(out[0], P["sam_mask_decoder.conv_s0.weight"], P["sam_mask_decoder.conv_s0.bias"])
    out[1] = F.conv2d(out[1], P["sam_mask_decoder.conv_s1.weight"], P["sam_mask_decoder.conv_s1.bias"])
    return {"backbone_fpn": out, "vision_pos_enc": pos}


def prepare_backbone_features(bo, n_levels=3):
    """S:2804-2818: NCHW -> (HW)NC for the last n levels."""
    fm, pe = bo["backbone_fpn"][-n_levels:], bo["vision_pos_enc"][-n_levels:]
    sizes = [(x.shape[-2], x.shape[-1]) for x in pe]
    return [x.flatten(2).permute(2, 0, 1) for x in fm], [x.flatten(2).permute(2, 0, 1) for x in pe], sizes


# ------------------------------------------------------------------------------------------------ SAM heads
def attention(q, k, v, P, pre, heads):
    """S:1457-1481"""
    q, k, v = lin(q, P, pre + "q_proj"), lin(k, P, pre + "k_proj"), lin(v, P, pre + "v_proj")
    sep = lambda t: t.reshape(t.shape[0], t.shape[1], heads, -1).transpose(1, 2)
    o = sdpa(sep(q), sep(k), sep(v)).transpose(1, 2)
    return lin(o.reshape(o.shape[0], o.shape[1], -1), P, pre + "out_proj")


def two_way_transformer(src, pos_src, tokens, P, cfg: Sam2Cfg, pre="sam_mask_decoder.transformer."):
    """S:1292-1336 + S:1383-1414"""
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos_src.flatten(2).permute(0, 2, 1)
    queries, query_pe = tokens, tokens
    H = cfg.dec_heads
    for i in range(cfg.dec_depth):
        p = f"{pre}layers.{i}."
        if i == 0:
            queries = attention(queries, queries, queries, P, p + "self_attn.", H)
        else:
            q = queries + query_pe
            queries = queries + attention(q, q, queries, P, p + "self_attn.", H)
        queries = layer_norm(queries, P, p + "norm1", 1e-5)
        q, k = queries + query_pe, keys + key_pe
        queries = layer_norm(queries + attention(q, k, keys, P, p + "cross_attn_token_to_image.", H), P, p + "norm2", 1e-5)
        queries = layer_norm(queries + mlp(queries, P, p + "mlp", 2), P, p + "norm3", 1e-5)
        q, k = queries + query_pe, keys + key_pe
        keys = layer_norm(keys + attention(k, q, queries, P, p + "cross_attn_image_to_token.", H), P, p + "norm4", 1e-5)
    q, k = queries + query_pe, keys + key_pe
    queries = queries + attention(q, k, keys, P, pre + "final_attn_token_to_image.", H)
    return layer_norm(queries, P, pre + "norm_final_attn", 1e-5), keys


def conv_transpose_2x2(x, w, b):
    return F.conv_transpose2d(x, w, b, stride=2)


def mask_decoder(P, image_embeddings, image_pe, sparse, dense, high_res_features, cfg: Sam2Cfg, multimask_output=True, internals=None):
    """S:2021-2160 MaskDecoder.forward/predict_masks (pred_obj_scores, high-res feats, multimask token for obj ptr)."""
    pre = "sam_mask_decoder."
    out_tokens = torch.cat([P[pre + "obj_score_token.weight"], P[pre + "iou_token.weight"], P[pre + "mask_tokens.weight"]], dim=0)
    tokens = torch.cat((out_tokens[None].expand(sparse.shape[0], -1, -1), sparse), dim=1)
    src = image_embeddings + dense
    pos_src = image_pe.repeat_interleave(tokens.shape[0], dim=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(src, pos_src, tokens, P, cfg)
    iou_tok = hs[:, 1]
    mask_toks = hs[:, 2:6]
    src = src.transpose(1, 2).view(b, c, h, w)
    feat_s0, feat_s1 = high_res_features
    up = conv_transpose_2x2(src, P[pre + "output_upscaling.0.weight"], P[pre + "output_upscaling.0.bias"]) + feat_s1
    up = F.gelu(layer_norm_2d(up, P, pre + "output_upscaling.1"))
    up = F.gelu(conv_transpose_2x2(up, P[pre + "output_upscaling.3.weight"], P[pre + "output_upscaling.3.bias"]) + feat_s0)
    hyper = torch.stack([mlp(mask_toks[:, i], P, f"{pre}output_hypernetworks_mlps.{i}", 3) for i in range(4)], dim=1)
    masks = (hyper @ up.view(up.shape[0], up.shape[1], -1)).view(up.shape[0], -1, up.shape[2], up.shape[3])
    if internals is not None:   # fixture construction (tests/golden/blobfit.py) and gradient localisation (tools/grad_locate.py): intermediates of the head
        internals.update(mask_toks=mask_toks, upscaled=up, hs=hs, src=src, tokens=tokens, hyper=hyper, masks=masks)
        for t in (hs, src, up, hyper, masks, tokens):
            if t.requires_grad:
                t.retain_grad()
    iou = mlp(iou_tok, P, pre + "iou_prediction_head", 3, sigmoid_output=True)
    obj_logits = mlp(hs[:, 0], P, pre + "pred_obj_score_head", 3)
    if multimask_output:
        return masks[:, 1:], iou[:, 1:], mask_toks[:, 1:], obj_logits
    return masks[:, 0:1], iou[:, 0:1], mask_toks[:, 0:1], obj_logits


def prompt_encoder_no_points(P, B, cfg: Sam2Cfg):
    """S:1674-1716 with points = one (0,0) label -1 (S:3322-3325) -> padded to two not-a-point rows; no mask."""
    pre = "sam_prompt_encoder."
    sparse = P[pre + "not_a_point_embed.weight"].expand(2, -1)[None].expand(B, -1, -1)
    s = cfg.image_size // cfg.backbone_stride
    dense = P[pre + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, s, s)
    dense_pe = position_embedding_random(P[pre + "pe_layer.positional_encoding_gaussian_matrix"], s, s)[None]
    return sparse, dense, dense_pe


def forward_sam_heads(P, backbone_features, high_res_features, language_embd, cfg: Sam2Cfg, multimask_output=True, internals=None):
    """S:3262-3431 (language path: no points, no mask prompt)."""
    B = backbone_features.shape[0]
    sparse, dense, dense_pe = prompt_encoder_no_points(P, B, cfg)
    if language_embd is not None:
        sparse = torch.cat([sparse, language_embd], dim=1)
    low_multi, ious, toks, obj_logits = mask_decoder(P, backbone_features, dense_pe, sparse, dense, high_res_features, cfg, multimask_output, internals)
    is_obj = obj_logits > 0
    low_multi = low_multi.float()
    high_multi = F.interpolate(low_multi, size=(cfg.image_size, cfg.image_size), mode="bilinear", align_corners=False)
    tok = toks[:, 0]
    best = torch.zeros(B, dtype=torch.long)
    if multimask_output:
        best = torch.argmax(ious, dim=-1)
        bi = torch.arange(B)
        low, high = low_multi[bi, best].unsqueeze(1), high_multi[bi, best].unsqueeze(1)
        tok = toks[bi, best]
    else:
        low, high = low_multi, high_multi
    obj_ptr = mlp(tok, P, "obj_ptr_proj", 3)
    lam = is_obj.float()
    obj_ptr = lam * obj_ptr + (1 - lam) * P["no_obj_ptr"]
    return dict(low_res_multimasks=low_multi, high_res_multimasks=high_multi, ious=ious, low_res_masks=low, high_res_masks=high,
                obj_ptr=obj_ptr, object_score_logits=obj_logits, best_iou_inds=best)


def inject_language_embd_train(P, feats, language_embd, cfg: Sam2Cfg, internals=None):
    """S:343-375: frames independent, + no_mem_embed, multimask always on (S:3128-3136)."""
    vf, _, sizes = feats
    B = vf[-1].shape[1]
    high = [x.permute(1, 2, 0).view(x.shape[1], x.shape[2], *s) for x, s in zip(vf[:-1], sizes[:-1])]
    pix = (vf[-1] + P["no_mem_embed"]).permute(1, 2, 0).view(B, cfg.d_model, *sizes[-1])
    o = forward_sam_heads(P, pix, high, language_embd, cfg, True, internals)
    return o["low_res_masks"], o["high_res_masks"], o


# ------------------------------------------------------------------------------------------------ memory
def memory_encoder(P, pix_feat, mask_for_mem, cfg: Sam2Cfg):
    """S:744-767 (skip_mask_sigmoid=True) + S:602-643 MaskDownSampler + S:690-703 CXBlock."""
    pre = "memory_encoder."
    x = mask_for_mem
    n_down = int(math.log2(cfg.backbone_stride))
    for i in range(n_down):
        x = F.conv2d(x, P[f"{pre}mask_downsampler.encoder.{3 * i}.weight"], P[f"{pre}mask_downsampler.encoder.{3 * i}.bias"], stride=2, padding=1)
        x = F.gelu(layer_norm_2d(x, P, f"{pre}mask_downsampler.encoder.{3 * i + 1}"))
    x = F.conv2d(x, P[f"{pre}mask_downsampler.encoder.{3 * n_down}.weight"], P[f"{pre}mask_downsampler.encoder.{3 * n_down}.bias"])
    y = F.conv2d(pix_feat, P[pre + "pix_feat_proj.weight"], P[pre + "pix_feat_proj.bias"]) + x
    for li in range(cfg.fuser_layers):
        p = f"{pre}fuser.layers.{li}."
        z = F.conv2d(y, P[p + "dwconv.weight"], P[p + "dwconv.bias"], padding=3, groups=y.shape[1])
        z = layer_norm_2d(z, P, p + "norm").permute(0, 2, 3, 1)
        z = lin(F.gelu(lin(z, P, p + "pwconv1")), P, p + "pwconv2") * P[p + "g_weight"]
        y = y + z.permute(0, 3, 1, 2)
    y = F.conv2d(y, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])
    pos = position_embedding_sine(cfg.mem_dim, y.shape[-2], y.shape[-1])[None].repeat(y.shape[0], 1, 1, 1)
    return y, pos


def encode_new_memory(P, vision_feats, sizes, high_res_masks, cfg: Sam2Cfg):
    """S:2991-3029"""
    B = vision_feats[-1].shape[1]
    pix = vision_feats[-1].permute(1, 2, 0).view(B, cfg.d_model, *sizes[-1])
    m = torch.sigmoid(high_res_masks) * cfg.sigmoid_scale_for_mem_enc + cfg.sigmoid_bias_for_mem_enc
    return memory_encoder(P, pix, m, cfg)


def rope_attention(q, k, v, P, pre, num_k_exclude_rope, rope_k_repeat):
    """S:1506-1548 (1 head).  Axial table recomputed from the query count (S:1520-1523)."""
    q, k, v = lin(q, P, pre + "q_proj"), lin(k, P, pre + "k_proj"), lin(v, P, pre + "v_proj")
    q, k, v = q[:, None], k[:, None], v[:, None]
    side = int(math.sqrt(q.shape[-2]))
    cos, sin = compute_axial_cis(q.shape[-1], side, side)
    n_rope = k.shape[-2] - num_k_exclude_rope
    qr, kr = apply_rotary_enc(q, k[:, :, :n_rope], cos, sin, repeat_freqs_k=rope_k_repeat)
    k = torch.cat([kr, k[:, :, n_rope:]], dim=2)
    o = sdpa(qr, k, v)[:, 0]
    return lin(o, P, pre + "out_proj")


def memory_attention(P, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens, cfg: Sam2Cfg):
    """S:550-600 + S:489-530. Inputs seq-first (N, B, C); returns (N, B, C)."""
    x = (curr + 0.1 * curr_pos).transpose(0, 1)
    mem, mpos = memory.transpose(0, 1), memory_pos.transpose(0, 1)
    for li in range(cfg.memattn_layers):
        p = f"memory_attention.layers.{li}."
        t = layer_norm(x, P, p + "norm1", 1e-5)
        x = x + rope_attention(t, t, t, P, p + "self_attn.", 0, False)
        t = layer_norm(x, P, p + "norm2", 1e-5)
        x = x + rope_attention(t, mem + mpos, mem, P, p + "cross_attn_image.", num_obj_ptr_tokens, True)
        t = layer_norm(x, P, p + "norm3", 1e-5)
        x = x + lin(F.relu(lin(t, P, p + "linear1")), P, p + "linear2")
    return layer_norm(x, P, "memory_attention.norm", 1e-5).transpose(0, 1)


def prepare_memory_conditioned_features(P, frame_idx, is_init_cond_frame, vf, vpos, sizes, output_dict, num_frames, cfg: Sam2Cfg, track_in_reverse=False):
    """S:2820-2989 (stride r=1, eval: only pointers on the already-tracked side, no tpos on pointers).  track_in_reverse (S:2829, :2866-2893, :2927-2944): the
    "previous" frames are frame_idx + t_rel -- whatever pass left a memory there -- and pointers come from frames t >= frame_idx."""
    B = vf[-1].shape[1]
    C, (H, W) = cfg.d_model, sizes[-1]
    if is_init_cond_frame:
        return (vf[-1] + P["no_mem_embed"]).permute(1, 2, 0).view(B, C, H, W)
    to_cat, to_cat_pos = [], []
    cond = output_dict["cond_frame_outputs"]
    sel, unsel = select_closest_cond_frames(frame_idx, cond, -1)
    prevs = [(0, o) for o in sel.values()]
    for t_pos in range(1, cfg.num_maskmem):
        t_rel = cfg.num_maskmem - t_pos
        prev_idx = frame_idx + t_rel if track_in_reverse else frame_idx - t_rel
        out = output_dict["non_cond_frame_outputs"].get(prev_idx, None)
        if out is None:
            out = unsel.get(prev_idx, None)
        prevs.append((t_pos, out))
    for t_pos, prev in prevs:
        if prev is None:
            continue
        to_cat.append(prev["maskmem_features"].float().flatten(2).permute(2, 0, 1))
        enc = prev["maskmem_pos_enc"].flatten(2).permute(2, 0, 1)
        to_cat_pos.append(enc + P["maskmem_tpos_enc"][cfg.num_maskmem - t_pos - 1])
    max_ptrs = min(num_frames, cfg.max_obj_ptrs_in_encoder)
    ptrs = [(abs(frame_idx - t), o["obj_ptr"]) for t, o in sel.items() if (t >= frame_idx if track_in_reverse else t <= frame_idx)]
    for t_diff in range(1, max_ptrs):
        t = frame_idx + t_diff if track_in_reverse else frame_idx - t_diff
        if t < 0 or t >= num_frames:
            break
        o = output_dict["non_cond_frame_outputs"].get(t, unsel.get(t, None))
        if o is not None:
            ptrs.append((t_diff, o["obj_ptr"]))
    n_ptr_tok = 0
    if ptrs:
        op = torch.stack([p for _, p in ptrs], dim=0)  # [n, B, C]
        opos = op.new_zeros(len(ptrs), B, cfg.mem_dim)
        if cfg.mem_dim < C:
            op = op.reshape(-1, B, C // cfg.mem_dim, cfg.mem_dim).permute(0, 2, 1, 3).flatten(0, 1)
            opos = opos.repeat_interleave(C // cfg.mem_dim, dim=0)
        to_cat.append(op)
        to_cat_pos.append(opos)
        n_ptr_tok = op.shape[0]
    memory, mpos = torch.cat(to_cat, dim=0), torch.cat(to_cat_pos, dim=0)
    out = memory_attention(P, vf[-1], vpos[-1], memory, mpos, n_ptr_tok, cfg)
    return out.permute(1, 2, 0).view(B, C, H, W)


def track_step(P, frame_idx, is_init_cond_frame, feats, output_dict, num_frames, cfg, run_mem_encoder, language_embd=None, track_in_reverse=False):
    """S:3160-3259"""
    vf, vpos, sizes = feats
    high = [x.permute(1, 2, 0).view(x.shape[1], x.shape[2], *s) for x, s in zip(vf[:-1], sizes[:-1])]
    pix = prepare_memory_conditioned_features(P, frame_idx, is_init_cond_frame, vf[-1:], vpos[-1:], sizes[-1:], output_dict, num_frames, cfg, track_in_reverse)
    # multimask: multimask_output_in_sam and (init or multimask_output_for_tracking) and 0 <= 0 pts <= 1 -> always True
    o = forward_sam_heads(P, pix, high, language_embd, cfg, True)
    cur = {"pred_masks": o["low_res_masks"], "pred_masks_high_res": o["high_res_masks"], "obj_ptr": o["obj_ptr"], "best_iou_inds": o["best_iou_inds"],
           "maskmem_features": None, "maskmem_pos_enc": None}
    if run_mem_encoder and cfg.num_maskmem > 0:
        mf, mp = encode_new_memory(P, vf, sizes, o["high_res_masks"], cfg)
        cur["maskmem_features"], cur["maskmem_pos_enc"] = mf.to(torch.bfloat16), mp  # stored bf16 (S:3607)
    return cur


class VideoSession:
    """Single-object restatement of the SAM2VideoPredictor state machine (S:3505-4132) as used by RGA3:
    add_language_embd on chosen frames (conditioning frames), then propagate_in_video over all frames."""

    def __init__(self, P, images, cfg: Sam2Cfg):
        self.P, self.images, self.cfg = P, images, cfg
        self.num_frames = images.shape[0]
        self.out = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        self.temp_cond: Dict[int, dict] = {}
        self.tracked = set()
        self.counts = {"enc": 0, "memattn": 0, "memenc": 0, "dec": 0}
        self._cache = None

    def _feats(self, t):
        if self._cache is None or self._cache[0] != t:  # one-entry cache (S:3539)
            bo = image_encoder_forward(self.P, self.images[t:t + 1].float(), self.cfg)
            self.counts["enc"] += 1
            self._cache = (t, prepare_backbone_features(bo))
        return self._cache[1]

    def add_language_embd(self, frame_idx, language_embd):
        """S:3824-3898: the frame becomes an initial conditioning frame; memory encoder deferred to preflight."""
        cur = track_step(self.P, frame_idx, True, self._feats(frame_idx), self.out, self.num_frames, self.cfg, False, language_embd)
        self.counts["dec"] += 1
        self.temp_cond[frame_idx] = cur
        return cur["pred_masks"]

    def _preflight(self):
        """S:3977-4047: consolidate (pred_masks at 1/4 res), re-encode memory from the bilinear-upsampled low-res mask."""
        for t, cur in sorted(self.temp_cond.items()):
            high = F.interpolate(cur["pred_masks"], size=(self.cfg.image_size,) * 2, mode="bilinear", align_corners=False)
            vf, _, sizes = self._feats(t)
            mf, mp = encode_new_memory(self.P, vf, sizes, high, self.cfg)
            self.counts["memenc"] += 1
            self.out["cond_frame_outputs"][t] = {"pred_masks": cur["pred_masks"], "obj_ptr": cur["obj_ptr"],
                                                 "maskmem_features": mf.to(torch.bfloat16), "maskmem_pos_enc": mp}
        self.temp_cond = {}

    def propagate(self, start_frame_idx=None, max_frame_num_to_track=None, reverse=False):
        """S:4049-4132: yields (frame_idx, video_res_masks [1,1,S,S]) in processing order."""
        self._preflight()
        start = min(self.out["cond_frame_outputs"]) if start_frame_idx is None else start_frame_idx
        n_track = self.num_frames if max_frame_num_to_track is None else max_frame_num_to_track
        if reverse:     # S:4085-4090
            order = range(start, max(start - n_track, 0) - 1, -1) if start > 0 else []
        else:           # S:4091-4095
            order = range(start, min(start + n_track, self.num_frames - 1) + 1)
        res = []
        for t in order:
            if t in self.out["cond_frame_outputs"]:
                pm = self.out["cond_frame_outputs"][t]["pred_masks"]
            else:
                feats = self._feats(t)
                cur = track_step(self.P, t, False, feats, self.out, self.num_frames, self.cfg, True, track_in_reverse=reverse)
                self.counts["memattn"] += 1
                self.counts["dec"] += 1
                self.counts["memenc"] += 1
                self.out["non_cond_frame_outputs"][t] = cur
                pm = cur["pred_masks"]
            self.tracked.add(t)
            res.append((t, F.interpolate(pm, size=(self.cfg.image_size,) * 2, mode="bilinear", align_corners=False)))
        return res


class MultiObjectSession:
    """n_obj objects over one clip (S:3771-4132 with len(obj_ids) > 1).  The reference keeps one output dictionary per object (S:3824-3898: add_language_embd runs
    _run_single_frame_inference with batch_size = 1 on that object's dictionary) and tracks all objects as ONE batch over shared image features (S:4049-4132 ->
    _run_single_frame_inference with batch_size = n_obj, S:3977-4047 consolidation).  With non_overlap_masks, non_overlap_masks_for_mem_enc and
    clear_non_cond_mem_* at their defaults (False: S:2392, :3512-3517) nothing couples the batch entries: every per-sample operation (memory attention, decoder,
    memory encoder) sees one object, so the batch is restated as one VideoSession per object and each yield concatenates them: [n_obj, 1, S, S]."""

    def __init__(self, P, images, cfg: Sam2Cfg, n_obj: int):
        self.sessions = [VideoSession(P, images, cfg) for _ in range(n_obj)]

    def add_language_embd(self, frame_idx, obj_idx, language_embd):
        return self.sessions[obj_idx].add_language_embd(frame_idx, language_embd)

    def propagate(self, **kw):
        per_obj = [s.propagate(**kw) for s in self.sessions]
        return [(per_obj[0][i][0], torch.cat([r[i][1] for r in per_obj], dim=0)) for i in range(len(per_obj[0]))]


def language_embd_inference(P, images, language_embd_per_frame, cfg: Sam2Cfg):
    """S:378-404: prompt EVERY object on EVERY frame, then propagate (which finds every frame consolidated).  language_embd_per_frame[t] is [n_obj, C] (or [C] / [1, C]
    for one object); returns ([T * n_obj, 1, S, S] -- the per-frame yields [n_obj, 1, S, S] concatenated on dim 0, frame-major, S:399-403 -- and the session)."""
    embs = [e.reshape(-1, e.shape[-1]) for e in language_embd_per_frame]
    n_obj = embs[0].shape[0]
    if n_obj == 1:
        sess = VideoSession(P, images, cfg)
        for t, e in enumerate(embs):
            sess.add_language_embd(t, e.reshape(1, 1, -1))
        return torch.cat([m for _, m in sess.propagate()], dim=0), sess
    sess = MultiObjectSession(P, images, cfg, n_obj)
    for t, e in enumerate(embs):
        for o in range(n_obj):
            sess.add_language_embd(t, o, e[o].reshape(1, 1, -1))
    return torch.cat([m for _, m in sess.propagate()], dim=0), sess
