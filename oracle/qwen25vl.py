"""ORACLE — test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product path).

fp32 PyTorch-CPU restatement of the Qwen2.5-VL arithmetic that RGA3's UniGRModel inherits
(reference model/qwen_2_5_vl_sam2.py:104 subclasses transformers' Qwen2_5_VLForConditionalGeneration; the
arithmetic itself is NOT vendored under /root/reference).  Third-party dependency: transformers, pinned by
the reference at 4.49.0.dev0 (requirements.txt:26) — not installable offline.  PARITY PIN: this file is
checked against golden vectors generated from the installed transformers 5.15.0 copy of the same module
(tests/golden/qwen_*.npz, made by tests/golden/make_qwen_fixtures.py); for the pinned 4.49 release itself the
parity is UNPINNED (no reference test or vector exists) — see DESIGN.md.

Line citations "HF:n" are to transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py (5.15.0) and "VU:n" to
transformers/vision_utils.py.  Parameter names follow the 4.49 checkpoint layout the released UniGR weights use
(visual.*, model.*, lm_head.*).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class VisionCfg:
    depth: int = 32
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 3420
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: Sequence[int] = (7, 15, 23, 31)
    out_hidden_size: int = 3584
    in_channels: int = 3
    tokens_per_second: int = 2


@dataclass
class TextCfg:
    hidden_size: int = 3584
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    intermediate_size: int = 18944
    vocab_size: int = 152064
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: Sequence[int] = (16, 24, 24)


@dataclass
class QwenCfg:
    vision: VisionCfg = field(default_factory=VisionCfg)
    text: TextCfg = field(default_factory=TextCfg)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652


# --------------------------------------------------------------------------------------------------------------
# integer plumbing (bit-exact targets)
# --------------------------------------------------------------------------------------------------------------
def vision_cu_seqlens(grid_thw) -> np.ndarray:
    """VU:42-65: one attention segment per temporal slice (h*w tokens)."""
    out = [0]
    for t, h, w in np.asarray(grid_thw).tolist():
        for _ in range(t):
            out.append(out[-1] + h * w)
    return np.asarray(out, dtype=np.int32)


def vision_position_ids(grid_thw, merge: int) -> np.ndarray:
    """VU:81-127: (h, w) ids laid out block-major over merge x merge blocks, repeated t times."""
    rows = []
    for t, h, w in np.asarray(grid_thw).tolist():
        ids = np.zeros((h // merge, w // merge, merge, merge, 2), dtype=np.int64)
        for bh in range(h // merge):
            for bw in range(w // merge):
                for ih in range(merge):
                    for iw in range(merge):
                        ids[bh, bw, ih, iw] = (bh * merge + ih, bw * merge + iw)
        rows.append(np.tile(ids.reshape(-1, 2), (t, 1)))
    return np.concatenate(rows, 0)


def vision_window_index(grid_thw, merge: int, window_size: int, patch_size: int):
    """VU:130-188: window reorder at merge-unit granularity + cumulative window lengths (in patches).

    Windows are vit_merger_window_size x vit_merger_window_size merged tokens; partial windows at the right /
    bottom edge are shorter; when the grid divides exactly, the reference's padding adds a fully empty window
    row/column whose zero length is dropped by unique_consecutive (VU:166-187) — reproduced here by skipping
    empty windows.
    """
    ws = window_size // merge // patch_size
    unit = merge * merge
    index: List[int] = []
    cu = [0]
    base = 0
    for t, h, w in np.asarray(grid_thw).tolist():
        lh, lw = h // merge, w // merge
        pad_h, pad_w = ws - lh % ws, ws - lw % ws
        nwh, nww = (lh + pad_h) // ws, (lw + pad_w) // ws
        for ti in range(t):
            for wh in range(nwh):
                for ww in range(nww):
                    n = 0
                    for ih in range(ws):
                        for iw in range(ws):
                            y, x = wh * ws + ih, ww * ws + iw
                            if y < lh and x < lw:
                                index.append(base + ti * lh * lw + y * lw + x)
                                n += 1
                    nxt = cu[-1] + n * unit
                    if nxt != cu[-1]:  # unique_consecutive
                        cu.append(nxt)
        base += t * lh * lw
    return np.asarray(index, dtype=np.int64), np.asarray(cu, dtype=np.int32)


def rope_index(input_ids, cfg: QwenCfg, image_grid_thw=None, video_grid_thw=None, second_per_grid_ts=None,
               attention_mask=None, temporal_rule: str = "hf449"):
    """HF:944-1058 (5.15) / get_rope_index of 4.49: 3-axis position ids for text + vision tokens.

    Runs of image/video placeholder tokens take (t, h, w) grid positions offset by the running position; text runs
    take 1-D positions.  temporal_rule: "hf515" -> time_interval = tokens_per_second * int(second_per_grid_t)
    (HF:1027); "hf449" -> index t uses int(t * second_per_grid_t * tokens_per_second) (the pinned release's
    floating form).  Both coincide for integer second_per_grid_ts.
    """
    ids = np.asarray(input_ids)
    B, S = ids.shape
    merge = cfg.vision.spatial_merge_size
    pos = np.zeros((3, B, S), dtype=np.int64)
    deltas = []
    img_it = iter(np.asarray(image_grid_thw).tolist()) if image_grid_thw is not None else iter(())
    vid_it = iter(np.asarray(video_grid_thw).tolist()) if video_grid_thw is not None else iter(())
    sec_it = iter(np.asarray(second_per_grid_ts).tolist()) if second_per_grid_ts is not None else None
    for b in range(B):
        keep = np.ones(S, dtype=bool) if attention_mask is None else np.asarray(attention_mask)[b].astype(bool)
        cur = ids[b][keep]
        types = np.where(cur == cfg.image_token_id, 1, np.where(cur == cfg.video_token_id, 2, 0))
        out = np.zeros((3, len(cur)), dtype=np.int64)
        i, cur_pos = 0, 0
        while i < len(cur):
            j = i
            while j < len(cur) and types[j] == types[i]:
                j += 1
            if types[i] == 0:
                out[:, i:j] = np.arange(j - i)[None, :] + cur_pos
                cur_pos += j - i
                i = j
                continue
            # a run of placeholder tokens may hold several images back to back only if separated by text
            # in practice (vision_start/end tokens); one grid per run as in HF:1016-1034.
            t, h, w = next(img_it) if types[i] == 1 else next(vid_it)
            lt, lh, lw = t, h // merge, w // merge
            n = lt * lh * lw
            tt = np.arange(lt)
            if types[i] == 2:
                spg = next(sec_it) if sec_it is not None else 1
                if temporal_rule == "hf515":
                    tt = tt * (cfg.vision.tokens_per_second * int(spg))
                else:
                    tt = (tt * float(spg) * cfg.vision.tokens_per_second).astype(np.int64)
            T, H, W = np.meshgrid(tt, np.arange(lh), np.arange(lw), indexing="ij")
            out[0, i:i + n] = T.reshape(-1) + cur_pos
            out[1, i:i + n] = H.reshape(-1) + cur_pos
            out[2, i:i + n] = W.reshape(-1) + cur_pos
            cur_pos += max(lh, lw) if temporal_rule == "hf515" else int(max(out[:, i:i + n].max() + 1 - cur_pos, 0))
            i += n
        pos[:, b, keep] = out
        deltas.append(out.max() + 1 - len(cur))
    return pos, np.asarray(deltas, dtype=np.int64)[:, None]


# --------------------------------------------------------------------------------------------------------------
# float arithmetic (fp32)
# --------------------------------------------------------------------------------------------------------------
# The reference RUNS this model in bf16 (app.py:53-58 torch_dtype=torch.bfloat16 / model.bfloat16(); train_joint.py:165-179 precision bf16): every module output is
# rounded to bf16 storage while the arithmetic inside an op is fp32.  `with storage(torch.bfloat16):` makes this restatement round at the same points (marked _r
# below: nn.Linear / conv outputs, both steps of the RMSNorm's cast-then-scale, rotary outputs, softmax probabilities and attention outputs as flash-attn / sdpa
# store them, activation outputs, residual sums, logits) -- the yardstick for full-depth comparisons: what bf16 STORAGE alone does to a 60-block stack, measured
# against the same code in fp32 (tests/test_fulldepth_parity_gpu.py).  Default: no rounding, plain fp32.
_STORE = [None]


def _r(x):
    return x if _STORE[0] is None else x.to(_STORE[0]).float()


class storage:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.old, _STORE[0] = _STORE[0], self.dtype

    def __exit__(self, *a):
        _STORE[0] = self.old


def _softmax_pv(scores, v):
    """softmax(scores) @ v.  fp32 mode: exactly that.  bf16-storage mode: as the fused attention kernels store it (flash-attn 2 on the reference's GPUs, torch's
    CPU flash kernel behind sdpa -- and this build's HIP kernels): the UN-normalised exponentials exp(s - rowmax) are rounded to bf16 for the second product, the row
    sums are taken from the fp32 exponentials, and the fp32 product is divided by them before the one rounding of the output.  Pinned against transformers' own bf16
    run by tests/golden/qwen_mid_bf16.npz (normalise-then-round, the eager form, sits 4 x further from that run)."""
    if _STORE[0] is None:
        return torch.nan_to_num(torch.softmax(scores, dim=-1)) @ v
    m = scores.amax(-1, keepdim=True)
    m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))     # fully masked (padding) query rows
    e = torch.exp(scores - m)
    return _r(torch.nan_to_num((_r(e) @ v) / e.sum(-1, keepdim=True)))


def rms_norm(x, w, eps):
    """HF:74-79 (the normalised rows are cast back to the input dtype BEFORE the weight multiplies them)"""
    v = x.pow(2).mean(-1, keepdim=True)
    return _r(w * _r(x * torch.rsqrt(v + eps)))


def rotate_half(x):
    """HF:153-157"""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _lin(x, P, name):
    """nn.Linear, plus the PEFT-style LoRA update when P carries <name>.lora_A/B.default.weight (scaling in P['lora_scaling'])."""
    y = _r(F.linear(x, P[name + ".weight"], P.get(name + ".bias")))
    a = P.get(name + ".lora_A.default.weight")
    if a is not None:
        xin = x
        dm = P.get("lora_dropout_masks")   # {<name>: (keep mask broadcastable to x, scale)}: PEFT's nn.Dropout on the lora_A input, with a
        if dm is not None and name in dm:  # GIVEN mask (tests pass the mask the build's counter hash produced)
            keep, scale = dm[name]
            xin = x * keep.to(x.dtype).reshape(x.shape) * scale
        y = y + P["lora_scaling"] * F.linear(F.linear(xin, a), P[name + ".lora_B.default.weight"])
    return y


def vit_forward(P: Dict[str, torch.Tensor], pixel_values: torch.Tensor, grid_thw, cfg: QwenCfg, return_pre_merge=False, trace: Optional[list] = None,
                blocks: Optional[Sequence[int]] = None, x_start: Optional[torch.Tensor] = None):
    """HF:408-471 Qwen2_5_VisionTransformerPretrainedModel.forward, fp32.

    pixel_values [N, C*Tp*P*P] (rows ordered t, h/2, w/2, 2, 2; cols C, Tp, P, P).
    Test hooks: trace collects the (window-ordered) residual stream after the patch embedding and after every block; blocks = the block indices to run (default all)
    starting from x_start [N, hidden] (window order) instead of the patch embedding -- one block on a GIVEN input (tests/test_oracle_qwen.py, teacher-forced pins)."""
    v = cfg.vision
    unit = v.spatial_merge_size ** 2
    hd = v.hidden_size // v.num_heads
    grid = np.asarray(grid_thw)
    pos_ids = torch.from_numpy(vision_position_ids(grid, v.spatial_merge_size))
    cu_full = vision_cu_seqlens(grid)
    win_idx_np, cu_win = vision_window_index(grid, v.spatial_merge_size, v.window_size, v.patch_size)
    win_idx = torch.from_numpy(win_idx_np)

    # patch embed: Conv3d with kernel == stride == (Tp, P, P), no bias == one matmul (HF:99-122)
    w = P["visual.patch_embed.proj.weight"].reshape(v.hidden_size, -1)
    x = _r(pixel_values.float() @ w.t())
    N = x.shape[0]
    x = x.reshape(N // unit, unit, -1)[win_idx].reshape(N, -1)  # HF:434-438
    if x_start is not None:
        x = x_start.float()
    if trace is not None:
        trace.append(x)

    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))  # HF:125-134, dim = hd/2
    rot = (pos_ids.unsqueeze(-1).float() * inv_freq).flatten(1)  # [N, hd/2]
    rot = rot.reshape(N // unit, unit, -1)[win_idx].reshape(N, -1)
    emb = torch.cat((rot, rot), dim=-1)
    cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]

    for li in (range(v.depth) if blocks is None else blocks):
        pre = f"visual.blocks.{li}."
        cu = cu_full if li in v.fullatt_block_indexes else cu_win
        h = rms_norm(x, P[pre + "norm1.weight"], 1e-6)
        qkv = _lin(h, P, pre + "attn.qkv").reshape(N, 3, v.num_heads, hd)
        q, k, val = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = _r(q * cos + rotate_half(q) * sin)  # HF:160-171 (fp32 inside, one cast back)
        k = _r(k * cos + rotate_half(k) * sin)
        att = torch.empty_like(q)
        lens = np.diff(cu)
        if len(lens) > 1 and (lens == lens[0]).all():  # equal-length segments: one batched product (same arithmetic)
            L = int(lens[0])
            qq, kk, vv = (z.reshape(-1, L, v.num_heads, hd).transpose(1, 2) for z in (q, k, val))
            att = _softmax_pv(qq @ kk.transpose(2, 3) * hd ** -0.5, vv).transpose(1, 2).reshape(N, v.num_heads, hd)
        else:
            for s in range(len(cu) - 1):  # HF:266-287: independent segments
                a, b = int(cu[s]), int(cu[s + 1])
                qq, kk, vv = (z[a:b].transpose(0, 1) for z in (q, k, val))
                att[a:b] = _softmax_pv(qq @ kk.transpose(1, 2) * hd ** -0.5, vv).transpose(0, 1)
        x = _r(x + _lin(att.reshape(N, -1), P, pre + "attn.proj"))
        h = rms_norm(x, P[pre + "norm2.weight"], 1e-6)
        h = _lin(_r(_r(F.silu(_lin(h, P, pre + "mlp.gate_proj"))) * _lin(h, P, pre + "mlp.up_proj")), P, pre + "mlp.down_proj")
        x = _r(x + h)
        if trace is not None:
            trace.append(x)
    pre_merge = x
    # merger (HF:137-150) then undo the window permutation (HF:464-466)
    h = rms_norm(x, P["visual.merger.ln_q.weight"], 1e-6).reshape(N // unit, -1)
    h = _lin(_r(F.gelu(_lin(h, P, "visual.merger.mlp.0"))), P, "visual.merger.mlp.2")
    out = h[torch.argsort(win_idx)]
    return (out, pre_merge) if return_pre_merge else out


def mrope_cos_sin(position_ids: torch.Tensor, cfg: QwenCfg):
    """HF:525-538 + :589-599: per-axis cos/sin then section interleave. position_ids [3,B,S] -> cos,sin [B,S,hd]."""
    t = cfg.text
    hd = t.hidden_size // t.num_attention_heads
    inv_freq = 1.0 / (t.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    freqs = position_ids[..., None].float() * inv_freq  # [3,B,S,hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    sec = list(t.mrope_section) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


def llm_forward(P, inputs_embeds, position_ids, attention_mask, cfg: QwenCfg, past=None, return_kv=False, hidden_states: Optional[list] = None):
    """HF:788-870 text model + :602-757 decoder layer, fp32, eager causal attention with key-padding mask.

    past: optional list of (k, v) [B,Hkv,Sp,hd] per layer (generate); returns post-final-norm hidden states."""
    t = cfg.text
    B, S, _ = inputs_embeds.shape
    hd = t.hidden_size // t.num_attention_heads
    rep = t.num_attention_heads // t.num_key_value_heads
    cos, sin = mrope_cos_sin(position_ids, cfg)
    cos, sin = _r(cos[:, None]), _r(sin[:, None])        # HF's rotary module returns cos / sin in the model dtype
    Sp = 0 if past is None else past[0][0].shape[2]
    i = torch.arange(S)[:, None] + Sp
    j = torch.arange(S + Sp)[None, :]
    mask = torch.zeros(B, 1, S, S + Sp)
    mask = mask.masked_fill(j > i, float("-inf"))
    if attention_mask is not None:
        am = attention_mask.bool()
        assert am.shape[1] == S + Sp
        mask = mask.masked_fill(~am[:, None, None, :], float("-inf"))
        # padded QUERY rows would be fully masked (NaN softmax / NaN gradients); let them see themselves — their outputs are
        # never read by valid rows (keys masked) nor by the loss (labels -100), so nothing observable changes
        eye = (j == i)[None, None]
        mask = torch.where(eye & ~am[:, None, -S:, None], torch.zeros_like(mask), mask)
    x = inputs_embeds.float()
    new_kv = []
    if hidden_states is not None:
        hidden_states.append(x)
    for li in range(t.num_hidden_layers):
        pre = f"model.layers.{li}."
        h = rms_norm(x, P[pre + "input_layernorm.weight"], t.rms_norm_eps)
        q = _lin(h, P, pre + "self_attn.q_proj").view(B, S, -1, hd).transpose(1, 2)
        k = _lin(h, P, pre + "self_attn.k_proj").view(B, S, -1, hd).transpose(1, 2)
        v = _lin(h, P, pre + "self_attn.v_proj").view(B, S, -1, hd).transpose(1, 2)
        q = _r(_r(q * cos) + _r(rotate_half(q) * sin))     # apply_multimodal_rotary_pos_emb works in the model dtype: each product and the sum are stored
        k = _r(_r(k * cos) + _r(rotate_half(k) * sin))
        if past is not None:
            k = torch.cat([past[li][0], k], dim=2)
            v = torch.cat([past[li][1], v], dim=2)
        new_kv.append((k, v))
        kk, vv = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
        sc = q @ kk.transpose(2, 3) * hd ** -0.5 + mask
        a = _softmax_pv(sc, vv).transpose(1, 2).reshape(B, S, -1)     # (fully masked padding query rows -> 0)
        x = _r(x + _lin(a, P, pre + "self_attn.o_proj"))
        h = rms_norm(x, P[pre + "post_attention_layernorm.weight"], t.rms_norm_eps)
        h = _lin(_r(_r(F.silu(_lin(h, P, pre + "mlp.gate_proj"))) * _lin(h, P, pre + "mlp.up_proj")), P, pre + "mlp.down_proj")
        x = _r(x + h)
        if hidden_states is not None:      # the residual stream after layer li (HF output_hidden_states[li + 1], the last one before the final norm)
            hidden_states.append(x)
    x = rms_norm(x, P["model.norm.weight"], t.rms_norm_eps)
    return (x, new_kv) if return_kv else x


def causal_lm_loss(logits, labels):
    """HF ForCausalLMLoss: shift, fp32 CE, mean over non-ignored (HF:1383-1393)."""
    lg = logits.float()[:, :-1].reshape(-1, logits.shape[-1])
    lb = labels[:, 1:].reshape(-1)
    return F.cross_entropy(lg, lb, ignore_index=-100, reduction="mean")


def forward(P, cfg: QwenCfg, input_ids, attention_mask=None, position_ids=None, labels=None, pixel_values=None,
            pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None, second_per_grid_ts=None,
            temporal_rule="hf449"):
    """HF:1185-1253 (Qwen2_5_VLModel.forward) + :1367-1402 (ForConditionalGeneration.forward).

    Returns dict(logits, loss, hidden (post final norm), position_ids)."""
    x = P["model.embed_tokens.weight"][input_ids]
    if pixel_values is not None:
        e = vit_forward(P, pixel_values, image_grid_thw, cfg)
        x = x.clone()
        x[input_ids == cfg.image_token_id] = e  # masked_scatter in row order (HF:1206-1223)
    if pixel_values_videos is not None:
        e = vit_forward(P, pixel_values_videos, video_grid_thw, cfg)
        x = x.clone()
        x[input_ids == cfg.video_token_id] = e
    if position_ids is None:
        pos, _ = rope_index(input_ids.numpy(), cfg, image_grid_thw, video_grid_thw, second_per_grid_ts,
                            None if attention_mask is None else attention_mask.numpy(), temporal_rule)
        position_ids = torch.from_numpy(pos)
    hidden = llm_forward(P, x, position_ids, attention_mask, cfg)
    logits = _r(hidden @ P["lm_head.weight"].t())
    loss = causal_lm_loss(logits, labels) if labels is not None else None
    return {"logits": logits, "loss": loss, "hidden": hidden, "position_ids": position_ids}
