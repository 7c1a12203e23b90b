"""Qwen2.5-VL host modules (ViT + decoder) running on the rga3 HIP kernels.

This is the build's own counterpart of the third-party model the reference subclasses
(reference model/qwen_2_5_vl_sam2.py:9-12,104 -> transformers Qwen2_5_VLForConditionalGeneration).  It keeps
the checkpoint parameter names of the pinned 4.49 layout (``visual.*``, ``model.*``, ``lm_head.*``) so released
UniGR weights load, and the substring-based LoRA / trainable selection of reference train_joint.py:199-251
keeps working (``q_proj``, ``v_proj``, ``lm_head``, ``embed_tokens``).

Arithmetic is delegated to hand-written HIP kernels through rga3.hip.ops; PyTorch provides tensors, streams
and autograd plumbing only.  There is no eager fallback: on a box without the extension every forward raises.

MI355X-first choices (vs. the reference's per-op eager graph):
  * tokens are PACKED (no pad rows); attention is varlen everywhere (ViT windows, causal decoder).
  * q/k/v projections are one GEMM into a [T, heads, D] buffer that RoPE rotates in place and the attention
    kernel reads through strides (no permute/contiguous copies).
  * gate/up projections are one GEMM whose epilogue applies SiLU(gate)*up (weights interleaved 16 rows gate /
    16 rows up at pack time); residual adds are fused into the proj / down GEMM epilogues.
  * ragged dims are padded once at weight-pack time (ViT MLP 3420 -> 3456, patch K 1176 -> 1216).
"""
from __future__ import annotations

import contextlib

import json
import math
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..hip import ops
from ..utils.staging import host_of, upload
from . import qwen_index as QI


# ------------------------------------------------------------------------------------------------ config
class Qwen2_5_VLVisionConfig:
    def __init__(self, depth=32, hidden_size=1280, num_heads=16, intermediate_size=3420, patch_size=14, temporal_patch_size=2,
                 spatial_merge_size=2, window_size=112, fullatt_block_indexes=(7, 15, 23, 31), out_hidden_size=3584, in_channels=3,
                 tokens_per_second=2, hidden_act="silu", **kwargs):
        self.depth, self.hidden_size, self.num_heads = depth, hidden_size, num_heads
        self.intermediate_size, self.patch_size, self.temporal_patch_size = intermediate_size, patch_size, temporal_patch_size
        self.spatial_merge_size, self.window_size = spatial_merge_size, window_size
        self.fullatt_block_indexes = list(fullatt_block_indexes)
        self.out_hidden_size, self.in_channels, self.tokens_per_second = out_hidden_size, in_channels, tokens_per_second
        self.hidden_act = hidden_act
        self.extra = kwargs

    def to_dict(self):
        d = {k: v for k, v in self.__dict__.items() if k != "extra"}
        d.update(self.extra)
        return d


class Qwen2_5_VLConfig:
    """Accepts the flat 4.49 config.json layout (hidden_size, ... + vision_config) and the nested 5.x one (text_config)."""
    model_type = "qwen2_5_vl"

    def __init__(self, vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                 num_key_value_heads=4, rms_norm_eps=1e-6, rope_theta=1000000.0, rope_scaling=None, vision_config=None,
                 image_token_id=151655, video_token_id=151656, vision_start_token_id=151652, vision_end_token_id=151653,
                 bos_token_id=151643, eos_token_id=151645, pad_token_id=None, tie_word_embeddings=False, text_config=None,
                 torch_dtype="bfloat16", **kwargs):
        if text_config is not None:  # 5.x nested layout
            tc = dict(text_config)
            vocab_size = tc.get("vocab_size", vocab_size)
            hidden_size = tc.get("hidden_size", hidden_size)
            intermediate_size = tc.get("intermediate_size", intermediate_size)
            num_hidden_layers = tc.get("num_hidden_layers", num_hidden_layers)
            num_attention_heads = tc.get("num_attention_heads", num_attention_heads)
            num_key_value_heads = tc.get("num_key_value_heads", num_key_value_heads)
            rms_norm_eps = tc.get("rms_norm_eps", rms_norm_eps)
            rp = tc.get("rope_parameters") or tc.get("rope_scaling") or {}
            rope_theta = tc.get("rope_theta", rp.get("rope_theta", rope_theta))
            rope_scaling = rope_scaling or rp
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads, self.num_key_value_heads = num_hidden_layers, num_attention_heads, num_key_value_heads
        self.rms_norm_eps, self.rope_theta = rms_norm_eps, rope_theta
        self.rope_scaling = rope_scaling or {"type": "mrope", "mrope_section": [16, 24, 24]}
        if isinstance(vision_config, Qwen2_5_VLVisionConfig):
            self.vision_config = vision_config
        else:
            self.vision_config = Qwen2_5_VLVisionConfig(**(vision_config or {}))
        self.image_token_id, self.video_token_id = image_token_id, video_token_id
        self.vision_start_token_id, self.vision_end_token_id = vision_start_token_id, vision_end_token_id
        self.bos_token_id, self.eos_token_id, self.pad_token_id = bos_token_id, eos_token_id, pad_token_id
        self.tie_word_embeddings = tie_word_embeddings
        self.torch_dtype = torch_dtype
        # which release's temporal-position rule get_rope_index follows (see rga3.model.qwen_index.rope_index)
        self.mrope_temporal_rule = kwargs.pop("mrope_temporal_rule", "hf449")
        self.extra = kwargs

    @property
    def mrope_section(self):
        return list(self.rope_scaling.get("mrope_section", [16, 24, 24]))

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    def to_dict(self):
        d = {k: v for k, v in self.__dict__.items() if k not in ("extra", "vision_config")}
        d["vision_config"] = self.vision_config.to_dict()
        d["model_type"] = self.model_type
        d.update({k: v for k, v in self.extra.items() if _jsonable(v)})
        return d

    @classmethod
    def from_dict(cls, d):
        return cls(**{k: v for k, v in d.items() if k != "model_type"})

    @classmethod
    def from_pretrained(cls, path, **overrides):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d.update(overrides)
        return cls.from_dict(d)

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2)


def _jsonable(v):
    try:
        json.dumps(v)
        return True
    except TypeError:
        return False


@dataclass
class CausalLMOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[list] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    rope_deltas: Optional[torch.Tensor] = None

    def __getitem__(self, i):
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states) if v is not None)[i]


# ------------------------------------------------------------------------------------------------ building blocks
class Linear(nn.Linear):
    """nn.Linear whose forward is the MFMA GEMM.  Subclasses nn.Linear so PEFT / name-based selection see a Linear."""

    def forward(self, x: torch.Tensor, residual=None, act="none") -> torch.Tensor:
        shp = x.shape
        y = ops.gemm(x.reshape(-1, shp[-1]), self.weight, self.bias, residual=None if residual is None else residual.reshape(-1, self.out_features), act=act)
        return y.view(*shp[:-1], y.shape[-1])


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.variance_epsilon = eps

    def forward(self, x):
        shp = x.shape
        return ops.rmsnorm(x.reshape(-1, shp[-1]), self.weight, self.variance_epsilon).view(shp)


def _is_plain(*mods):
    """True if the projections are this file's un-wrapped Linear (fused fast path allowed)."""
    return all(type(m) is Linear for m in mods)


def _versions(*params):
    return tuple((p.data_ptr(), p._version, p.dtype) for p in params if p is not None)


# RMSNorm folded into the products on either side of it (inference / frozen towers only): the residual-producing projection (o_proj, down_proj, the ViT's proj /
# fc2) leaves the row sums of squares of what it writes (rga3_gemm_rms_bf16, rms_out), the projection that consumes the normalised rows (q|k|v, gate|up) takes the
# UN-normalised rows with the norm weight folded into its weight and scales its accumulators by 1 / rms (rms_in).  The stand-alone norm launch and the write +
# re-read of the normalised rows disappear (2 x 57 launches of ~9 us in the decoder, 2 x 32 in the vision tower: profiles/r03_bench_forward_kernel_stats.csv).
# Rounding differs from the eager order by the bf16 rounding of gamma * W (instead of the normalised rows): inside the stated 2e-2.  RGA3_RMS_FOLD=0 switches it off.
_RMS_FOLD = os.environ.get("RGA3_RMS_FOLD", "1") != "0"
_ROPE_IN_ATTN = os.environ.get("RGA3_ROPE_IN_ATTN", "1") != "0"   # decoder prefill: queries rotated inside the causal attention kernel (inference route)


def rms_fold_enabled() -> bool:
    return _RMS_FOLD


def set_rms_fold(on: bool):
    """A/B switch (bench.py --no-rms-fold, tests)."""
    global _RMS_FOLD
    _RMS_FOLD = bool(on)


def set_rope_in_attn(on: bool):
    """A/B switch (tests): decoder prefill queries rotated inside the causal attention kernel instead of by the stand-alone RoPE pass."""
    global _ROPE_IN_ATTN
    _ROPE_IN_ATTN = bool(on)


def _fold_ok(x2d, *mods) -> bool:
    # (no gradient is recorded under no_grad whatever requires_grad says; the folded weight copies follow their sources' version counters)
    return _RMS_FOLD and not torch.is_grad_enabled() and x2d.shape[0] > 16 and x2d.dtype == torch.bfloat16 and _is_plain(*mods)


def _interleave_rows(g: torch.Tensor, u: torch.Tensor, pad_to: int) -> torch.Tensor:
    """[I, K] gate / up -> [2*Ip, K] with 16-row blocks alternating gate / up (GEMM SwiGLU epilogue layout)."""
    I = g.shape[0]
    if pad_to != I:
        z = torch.zeros((pad_to - I,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        g, u = torch.cat([g, z]), torch.cat([u, z])
    rest = tuple(g.shape[1:])
    return torch.stack([g.view(pad_to // 16, 16, *rest), u.view(pad_to // 16, 16, *rest)], dim=1).reshape(2 * pad_to, *rest).contiguous()


class GatedMLP(nn.Module):
    """down(silu(gate(x)) * up(x)) — HF Qwen2_5_VLMLP (modeling_qwen2_5_vl.py:84-96), Qwen2MLP (:541-553)."""

    def __init__(self, hidden, inter, bias):
        super().__init__()
        self.gate_proj = Linear(hidden, inter, bias=bias)
        self.up_proj = Linear(hidden, inter, bias=bias)
        self.down_proj = Linear(inter, hidden, bias=bias)
        self._pk = None

    def _packed(self):
        key = _versions(self.gate_proj.weight, self.up_proj.weight, self.down_proj.weight, self.gate_proj.bias, self.up_proj.bias)
        if self._pk is None or self._pk[0] != key:
            I = self.gate_proj.out_features
            Ip = (I + 63) // 64 * 64  # K of the down GEMM must be a multiple of 64
            with torch.no_grad():
                wgu = _interleave_rows(self.gate_proj.weight.detach(), self.up_proj.weight.detach(), Ip)
                bgu = None
                if self.gate_proj.bias is not None:
                    bgu = _interleave_rows(self.gate_proj.bias.detach(), self.up_proj.bias.detach(), Ip)
                wd = self.down_proj.weight.detach()
                if Ip != I:
                    wd = torch.cat([wd, torch.zeros((wd.shape[0], Ip - I), dtype=wd.dtype, device=wd.device)], dim=1).contiguous()
            self._pk = (key, wgu, bgu, wd)
        return self._pk[1:]

    def _packed_folded(self, norm):
        """gate|up pack with the preceding RMSNorm's weight folded in (W diag(gamma), one bf16 rounding), rebuilt when a source changes."""
        wgu, _, _ = self._packed()
        key = (self._pk[0], _versions(norm.weight))
        pf = self.__dict__.get("_pkf")
        if pf is None or pf[0] != key:
            with torch.no_grad():
                pf = (key, (wgu.float() * norm.weight.detach().float()[None, :]).to(wgu.dtype).contiguous())
            self.__dict__["_pkf"] = pf
        return pf[1]

    def forward(self, x2d, residual, fold=None, rms_out=None):
        """fold = (norm module, row sums of squares of x2d): x2d is then the UN-normalised row block (== residual) and the norm is applied inside the gate|up product."""
        if _is_plain(self.gate_proj, self.up_proj, self.down_proj):
            wgu, bgu, wd = self._packed()
            if fold is not None:
                norm, rs = fold
                a = ops.gemm(x2d, self._packed_folded(norm), bgu, act="swiglu", rms_in=(rs, x2d.shape[1], norm.variance_epsilon))
            else:
                a = ops.gemm(x2d, wgu, bgu, act="swiglu")
            return ops.gemm(a, wd, self.down_proj.bias, residual=residual, rms_out=rms_out)
        a = ops.silu_mul(self.gate_proj(x2d).contiguous(), self.up_proj(x2d).contiguous())
        return ops.add(self.down_proj(a).contiguous(), residual)


# ------------------------------------------------------------------------------------------------ vision tower
_VIT_BLOCK_FP8 = [None]      # set by rga3.model.qwen_train.set_fp8_frozen_gemms: the e4m3 forward of one frozen vision block (kept there with the other fp8 plumbing)


class VisionPatchEmbed(nn.Module):
    def __init__(self, c: Qwen2_5_VLVisionConfig):
        super().__init__()
        k = (c.temporal_patch_size, c.patch_size, c.patch_size)
        self.proj = nn.Conv3d(c.in_channels, c.hidden_size, kernel_size=k, stride=k, bias=False)  # parameter container only
        self._pk = None

    def forward(self, px):
        w = self.proj.weight
        key = _versions(w)
        if self._pk is None or self._pk[0] != key:
            w2 = w.detach().reshape(w.shape[0], -1)
            kp = (w2.shape[1] + 63) // 64 * 64
            self._pk = (key, ops.pad_cols(w2.contiguous(), kp) if kp != w2.shape[1] else w2.contiguous())
        wp = self._pk[1]
        if px.shape[1] != wp.shape[1]:
            px = ops.pad_cols(px, wp.shape[1])
        return ops.gemm(px, wp)


class VisionAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.num_heads = c.num_heads
        self.head_dim = c.hidden_size // c.num_heads
        self.qkv = Linear(c.hidden_size, c.hidden_size * 3, bias=True)
        self.proj = Linear(c.hidden_size, c.hidden_size, bias=True)

    def _qkv_folded(self, norm):
        key = _versions(self.qkv.weight, norm.weight)
        pf = self.__dict__.get("_pkf")
        if pf is None or pf[0] != key:
            with torch.no_grad():
                pf = (key, (self.qkv.weight.detach().float() * norm.weight.detach().float()[None, :]).to(self.qkv.weight.dtype).contiguous())
            self.__dict__["_pkf"] = pf
        return pf[1]

    def forward(self, h, residual, cu, max_len, cos, sin, fold=None, rms_out=None):
        N = h.shape[0]
        H, D = self.num_heads, self.head_dim
        if fold is not None:      # h is the un-normalised row block: norm1 is applied inside the qkv product
            norm, rs = fold
            qkv = ops.gemm(h, self._qkv_folded(norm), self.qkv.bias, rms_in=(rs, h.shape[1], norm.variance_epsilon)).view(N, 3 * H, D)
        else:
            qkv = self.qkv(h).view(N, 3 * H, D)
        if ops.attn_rope_win_ok(max_len, D) and not torch.is_grad_enabled():
            # windows: every key is loaded once per head, so q AND k are rotated while the attention kernel loads them -- no rope pass at all
            att = ops.attn_varlen_rope(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], cu, cu, max_len, D ** -0.5, cos, sin, causal=False, rope_k=True)
        else:
            ops.rope_(qkv, cos, sin, 0, 2 * H)  # q heads then k heads are contiguous in the fused buffer
            att = ops.attn_varlen(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], cu, cu, max_len, D ** -0.5, causal=False, max_k=max_len)
        if rms_out is not None:
            return ops.gemm(att.view(N, H * D), self.proj.weight, self.proj.bias, residual=residual, rms_out=rms_out)
        return self.proj(att.view(N, H * D), residual=residual)


class VisionBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm1 = RMSNorm(c.hidden_size, 1e-6)
        self.norm2 = RMSNorm(c.hidden_size, 1e-6)
        self.attn = VisionAttention(c)
        self.mlp = GatedMLP(c.hidden_size, c.intermediate_size, bias=True)

    def forward(self, x, cu, max_len, cos, sin, rs=None):
        """rs = (row sums of squares of x or None, buffer for this block's mid sums, buffer for the sums of this block's output or None): the RMSNorm-folded route."""
        if rs is None:
            x = self.attn(self.norm1(x), x, cu, max_len, cos, sin)
            return self.mlp(self.norm2(x), x)
        rs_x, rs_mid, rs_next = rs
        if rs_x is None:       # first block: nothing produced the sums of its input
            x = self.attn(self.norm1(x), x, cu, max_len, cos, sin, rms_out=rs_mid)
        else:
            x = self.attn(x, x, cu, max_len, cos, sin, fold=(self.norm1, rs_x), rms_out=rs_mid)
        return self.mlp(x, x, fold=(self.norm2, rs_mid), rms_out=rs_next)


class PatchMerger(nn.Module):
    def __init__(self, dim, context_dim, merge):
        super().__init__()
        self.hidden_size = context_dim * merge * merge
        self.ln_q = RMSNorm(context_dim, 1e-6)
        self.mlp = nn.Sequential(Linear(self.hidden_size, self.hidden_size), nn.GELU(), Linear(self.hidden_size, dim))

    def forward(self, x):
        h = self.ln_q(x).view(-1, self.hidden_size)
        h = self.mlp[0](h, act="gelu")
        return self.mlp[2](h)


class VisionTransformer(nn.Module):
    """HF Qwen2_5_VisionTransformerPretrainedModel (modeling_qwen2_5_vl.py:345-471) on HIP kernels."""

    def __init__(self, c: Qwen2_5_VLVisionConfig):
        super().__init__()
        self.config = c
        self.spatial_merge_size = c.spatial_merge_size
        self.patch_embed = VisionPatchEmbed(c)
        self.blocks = nn.ModuleList([VisionBlock(c) for _ in range(c.depth)])
        self.merger = PatchMerger(c.out_hidden_size, c.hidden_size, c.spatial_merge_size)
        self._plans = {}

    @property
    def dtype(self):
        return self.patch_embed.proj.weight.dtype

    def plan(self, grid_thw, device):
        """Index plan for a grid (cached: grids repeat across steps)."""
        c = self.config
        key = (tuple(map(tuple, np.asarray(grid_thw).reshape(-1, 3).tolist())), str(device))
        if key in self._plans:
            return self._plans[key]
        g = np.asarray(grid_thw).reshape(-1, 3)
        wi, cu_win = QI.vision_window_index(g, c.spatial_merge_size, c.window_size, c.patch_size)
        cu_full = QI.vision_cu_seqlens(g)
        pos = QI.vision_position_ids(g, c.spatial_merge_size)
        unit = c.spatial_merge_size ** 2
        hd = c.hidden_size // c.num_heads
        # rotary table in window order, fp32 (modeling_qwen2_5_vl.py:125-134, 441-446)
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32, device=device) / (hd // 2)))
        p = torch.from_numpy(pos).to(device)
        rot = (p.unsqueeze(-1).float() * inv).flatten(1)
        wi_d = torch.from_numpy(wi).to(device)
        rot = rot.view(-1, unit, rot.shape[-1])[wi_d].reshape(pos.shape[0], -1)
        emb = torch.cat((rot, rot), dim=-1)
        plan = dict(window_index=wi_d, cu_win=torch.from_numpy(cu_win).to(device), cu_full=torch.from_numpy(cu_full).to(device),
                    max_win=int(np.diff(cu_win).max()), max_full=int(np.diff(cu_full).max()),
                    cos=emb.cos().contiguous(), sin=emb.sin().contiguous(), n=int(pos.shape[0]))
        self._plans[key] = plan
        return plan

    def forward(self, pixel_values, grid_thw):
        c = self.config
        pl = self.plan(grid_thw, pixel_values.device)
        unit = c.spatial_merge_size ** 2
        x = self.patch_embed(pixel_values.to(self.dtype))
        x = ops.gather_rows(x, pl["window_index"], rows_per_idx=unit)
        plain = [m_ for b_ in self.blocks for m_ in (b_.attn.qkv, b_.attn.proj, b_.mlp.gate_proj, b_.mlp.up_proj, b_.mlp.down_proj)]
        hook = _VIT_BLOCK_FP8[0]
        if hook is not None and not torch.is_grad_enabled() and x.shape[1] % 128 == 0 and x.dtype == torch.bfloat16 and _is_plain(*plain):
            # BASELINE configs[4]: the frozen tower's contractions in e4m3 (rga3.model.qwen_train.vision_block_forward_fp8; norms, RoPE, attention stay bf16)
            for i, blk in enumerate(self.blocks):
                full = i in c.fullatt_block_indexes
                x = hook(blk, x, pl["cu_full" if full else "cu_win"], pl["max_full" if full else "max_win"], pl["cos"], pl["sin"])
            m = self.merger(x)
            out = torch.empty_like(m)
            ops.scatter_rows_(out, pl["window_index"], m)
            return out
        fold = _fold_ok(x, *plain)
        nb = len(self.blocks)
        sums = torch.zeros((2 * nb, x.shape[0]), dtype=torch.int64, device=x.device) if fold else None    # one memset for the whole tower
        for i, blk in enumerate(self.blocks):
            rs = (sums[2 * i - 1] if i > 0 else None, sums[2 * i], sums[2 * i + 1] if i + 1 < nb else None) if fold else None
            if i in c.fullatt_block_indexes:
                x = blk(x, pl["cu_full"], pl["max_full"], pl["cos"], pl["sin"], rs=rs)
            else:
                x = blk(x, pl["cu_win"], pl["max_win"], pl["cos"], pl["sin"], rs=rs)
        m = self.merger(x)
        out = torch.empty_like(m)
        ops.scatter_rows_(out, pl["window_index"], m)  # == merged[argsort(window_index)]
        return out


# ------------------------------------------------------------------------------------------------ decoder
class DecoderAttention(nn.Module):
    def __init__(self, c: Qwen2_5_VLConfig, layer_idx):
        super().__init__()
        self.layer_idx = layer_idx
        self.num_heads, self.num_kv, self.head_dim = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        hs = c.hidden_size
        self.q_proj = Linear(hs, self.num_heads * self.head_dim, bias=True)
        self.k_proj = Linear(hs, self.num_kv * self.head_dim, bias=True)
        self.v_proj = Linear(hs, self.num_kv * self.head_dim, bias=True)
        self.o_proj = Linear(self.num_heads * self.head_dim, hs, bias=False)
        self._pk = None

    def _packed(self):
        ps = (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias, self.k_proj.bias, self.v_proj.bias)
        key = _versions(*ps)
        if self._pk is None or self._pk[0] != key:
            with torch.no_grad():
                w = torch.cat([p.detach() for p in ps[:3]]).contiguous()
                b = torch.cat([p.detach() for p in ps[3:]]).contiguous()
            self._pk = (key, w, b)
        return self._pk[1:]

    def _packed_folded(self, norm):
        w, _ = self._packed()
        key = (self._pk[0], _versions(norm.weight))
        pf = self.__dict__.get("_pkf")
        if pf is None or pf[0] != key:
            with torch.no_grad():
                pf = (key, (w.float() * norm.weight.detach().float()[None, :]).to(w.dtype).contiguous())
            self.__dict__["_pkf"] = pf
        return pf[1]

    def forward(self, h, residual, cos, sin, cu, max_len, cache=None, fold=None, rms_out=None):
        T = h.shape[0]
        Hq, Hk, D = self.num_heads, self.num_kv, self.head_dim
        if fold is not None:      # h is the un-normalised row block: input_layernorm is applied inside the q|k|v product
            norm, rs = fold
            _, b = self._packed()
            qkv = ops.gemm(h, self._packed_folded(norm), b, rms_in=(rs, h.shape[1], norm.variance_epsilon)).view(T, Hq + 2 * Hk, D)
        elif _is_plain(self.q_proj, self.k_proj, self.v_proj):
            w, b = self._packed()
            qkv = ops.gemm(h, w, b).view(T, Hq + 2 * Hk, D)
        elif all(type(m).__name__ in ("Linear", "LoRALinear") and type(m).__module__.startswith("rga3.") for m in (self.q_proj, self.k_proj, self.v_proj)):
            from .qwen_train import qkv_with_lora  # native LoRA: fused projection + in-place low-rank updates
            qkv = qkv_with_lora(self, h, seeds=getattr(self, "_lora_drop_seeds", None))[0].view(T, Hq + 2 * Hk, D)
        else:  # foreign wrappers (e.g. PEFT): honour them, then assemble the fused buffer
            qkv = torch.cat([self.q_proj(h), self.k_proj(h), self.v_proj(h)], dim=-1).view(T, Hq + 2 * Hk, D)
        q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hk], qkv[:, Hq + Hk:]
        # prefill without autograd: only the Hk key heads go through the stand-alone RoPE pass, the queries are rotated by the attention kernel as it loads them
        # (rga3_attn_varlen_fwd_rope -> attn_causal32_kernel: the rotate-half partner d + 64 sits in the same lane)
        q_in_attn = _ROPE_IN_ATTN and not torch.is_grad_enabled() and D == 128 and max_len >= 256 and qkv.dtype == torch.bfloat16
        if q_in_attn:
            ops.rope_(qkv, cos, sin, Hq, Hk)
        else:
            ops.rope_(qkv, cos, sin, 0, Hq + Hk)
        cu_k = cu
        if cache is not None:
            k, v, cu_k = cache.update(self.layer_idx, k, v, cu)
        # decode step (one query per sequence): every cached key is visible, and the non-causal form may split the key range over workgroups
        if q_in_attn:
            att = ops.attn_varlen_rope(q, k, v, cu, cu_k, max_len, D ** -0.5, cos, sin, causal=True)
        else:
            att = ops.attn_varlen(q, k, v, cu, cu_k, max_len, D ** -0.5, causal=(max_len > 1))
        if rms_out is not None:
            return ops.gemm(att.view(T, Hq * D), self.o_proj.weight, self.o_proj.bias, residual=residual, rms_out=rms_out)
        return self.o_proj(att.view(T, Hq * D), residual=residual)


class DecoderLayer(nn.Module):
    def __init__(self, c, layer_idx):
        super().__init__()
        self.self_attn = DecoderAttention(c, layer_idx)
        self.mlp = GatedMLP(c.hidden_size, c.intermediate_size, bias=False)
        self.input_layernorm = RMSNorm(c.hidden_size, c.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(c.hidden_size, c.rms_norm_eps)

    def forward(self, x, cos, sin, cu, max_len, cache=None, rs=None):
        """rs = (sums of squares of x's rows or None, buffer for the post-attention sums, buffer for the sums of this layer's output or None): RMSNorm-folded route."""
        if rs is None:
            x = self.self_attn(self.input_layernorm(x), x, cos, sin, cu, max_len, cache)
            return self.mlp(self.post_attention_layernorm(x), x)
        rs_x, rs_mid, rs_next = rs
        if rs_x is None:       # first layer: the embeddings come without sums
            x = self.self_attn(self.input_layernorm(x), x, cos, sin, cu, max_len, cache, rms_out=rs_mid)
        else:
            x = self.self_attn(x, x, cos, sin, cu, max_len, cache, fold=(self.input_layernorm, rs_x), rms_out=rs_mid)
        return self.mlp(x, x, fold=(self.post_attention_layernorm, rs_mid), rms_out=rs_next)


class KVCache:
    """Packed per-sequence KV cache for generate(): one [B, cap, Hkv, D] buffer per layer and tensor."""

    def __init__(self, n_layers, batch, cap, hkv, d, device, dtype):
        self.k = [torch.empty((batch, cap, hkv, d), device=device, dtype=dtype) for _ in range(n_layers)]
        self.v = [torch.empty((batch, cap, hkv, d), device=device, dtype=dtype) for _ in range(n_layers)]
        self.lens = [0] * batch  # tokens stored per sequence (host)
        self.cap = cap
        self._new = None
        self._cu = None

    def get_seq_length(self):
        return max(self.lens) if self.lens else 0

    def begin(self, new_lens):
        self._new = list(new_lens)
        self._cu = None   # cu_k of this step: the same for every layer, built (one host -> device copy) by the first update()

    def update(self, layer, k, v, cu_q):
        """Append this step's packed k/v and return packed (k_all, v_all, cu_k) views for attention."""
        B = len(self.lens)
        off = 0
        ks, vs, cu = [], [], [0]
        for b in range(B):
            n = self._new[b]
            s = self.lens[b]
            self.k[layer][b, s:s + n].copy_(k[off:off + n])
            self.v[layer][b, s:s + n].copy_(v[off:off + n])
            off += n
            ks.append(self.k[layer][b, :s + n])
            vs.append(self.v[layer][b, :s + n])
            cu.append(cu[-1] + s + n)
        if B == 1:
            kk, vv = ks[0], vs[0]
        else:
            kk, vv = torch.cat(ks), torch.cat(vs)
        if self._cu is None:
            self._cu = torch.tensor(cu, dtype=torch.int32, device=k.device)
        return kk, vv, self._cu

    def commit(self):
        self.lens = [a + b for a, b in zip(self.lens, self._new)]
        self._new = None


class StaticKVCache:
    """Single-sequence cache view for the captured decode step: the same [1, cap, Hkv, D] buffers as KVCache, but the write position and the
    visible length live ON THE DEVICE (len_dev / cu_dev), so one captured graph serves every step: the new k / v rows go in by a row scatter at
    len_dev, attention sees the full-capacity buffer with cu_k = [0, len + 1], and advance() moves both inside the graph."""

    def __init__(self, base: "KVCache", device):
        assert len(base.lens) == 1
        self.k, self.v, self.cap = base.k, base.v, base.cap
        self.len_dev = torch.tensor([base.lens[0]], dtype=torch.int64, device=device)
        self.cu_dev = torch.tensor([0, base.lens[0] + 1], dtype=torch.int32, device=device)

    def reset(self, n_stored: int):
        self.len_dev.fill_(n_stored)
        self.cu_dev[1] = n_stored + 1

    def begin(self, new_lens):
        pass

    def update(self, layer, k, v, cu_q):
        ops.scatter_rows_(self.k[layer][0].view(self.cap, -1), self.len_dev, k.reshape(1, -1))
        ops.scatter_rows_(self.v[layer][0].view(self.cap, -1), self.len_dev, v.reshape(1, -1))
        return self.k[layer][0], self.v[layer][0], self.cu_dev

    def commit(self):
        pass

    def advance(self):
        self.len_dev += 1
        self.cu_dev[1:] += 1


class TextModel(nn.Module):
    def __init__(self, c: Qwen2_5_VLConfig):
        super().__init__()
        self.config = c
        self.embed_tokens = nn.Embedding(c.vocab_size, c.hidden_size)
        self.layers = nn.ModuleList([DecoderLayer(c, i) for i in range(c.num_hidden_layers)])
        self.norm = RMSNorm(c.hidden_size, c.rms_norm_eps)
        self._axis = None

    def mrope_tables(self, pos3):
        """pos3 [3, T] int64 (device) -> cos/sin [T, D] fp32 with the mrope section interleave
        (modeling_qwen2_5_vl.py:525-538, 589-599)."""
        c = self.config
        D = c.head_dim
        dev = pos3.device
        if self._axis is None or self._axis.device != dev:
            sec = c.mrope_section
            self._axis = torch.tensor(sum([[i] * s for i, s in enumerate(sec)], []), device=dev)
            self._inv = 1.0 / (c.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32, device=dev) / D))
        psel = pos3[self._axis, :].t().float()  # [T, D/2]: axis chosen per frequency index
        fr = psel * self._inv
        emb = torch.cat((fr, fr), dim=-1)
        return emb.cos().contiguous(), emb.sin().contiguous()

    def forward(self, x, pos3, cu, max_len, cache=None, collect_hidden=False):
        cos, sin = self.mrope_tables(pos3)
        hs = [x] if collect_hidden else None
        fold = _fold_ok(x, *[m_ for l_ in self.layers for m_ in (l_.self_attn.q_proj, l_.self_attn.k_proj, l_.self_attn.v_proj, l_.self_attn.o_proj,
                                                                  l_.mlp.gate_proj, l_.mlp.up_proj, l_.mlp.down_proj)])
        nl = len(self.layers)
        sums = torch.zeros((2 * nl, x.shape[0]), dtype=torch.int64, device=x.device) if fold else None    # one memset for all layers
        for i, layer in enumerate(self.layers):
            rs = (sums[2 * i - 1] if i > 0 else None, sums[2 * i], sums[2 * i + 1] if i + 1 < nl else None) if fold else None
            x = layer(x, cos, sin, cu, max_len, cache, rs=rs)
            if collect_hidden:
                hs.append(x)
        x = self.norm(x)
        if collect_hidden:
            hs[-1] = x  # HF replaces the last entry by the post-norm state
        return x, hs


class Qwen2_5_VLForConditionalGeneration(nn.Module):
    """Module tree and names of the pinned checkpoint layout: .visual, .model, .lm_head."""
    config_class = Qwen2_5_VLConfig

    def __init__(self, config: Qwen2_5_VLConfig):
        super().__init__()
        self.config = config
        self.visual = VisionTransformer(config.vision_config)
        self.model = TextModel(config)
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False)
        if config.tie_word_embeddings:   # Qwen2.5-VL-3B: the output embedding IS the input embedding (HF tie_weights)
            self.lm_head.weight = self.model.embed_tokens.weight
        self.rope_deltas = None
        self.gradient_checkpointing = False

    # -- HF surface used by the reference's callers --------------------------------------
    @property
    def device(self):
        return self.lm_head.weight.device

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def gradient_checkpointing_enable(self, *a, **k):
        """reference train_joint.py:188.  Wired to the decoder's activation-recompute switch: a layer then saves only its input and recomputes the rest in backward
        (qwen_train.set_activation_recompute) -- the reference's memory behaviour.  On a 288 GB part the kept activations (8 GB at S = 2112) fit: leave this call out,
        or call gradient_checkpointing_disable() after it, for the faster keep-activations step (same arithmetic, tests/test_train_gpu.py)."""
        from . import qwen_train
        self.gradient_checkpointing = True
        qwen_train.set_activation_recompute(True)

    def gradient_checkpointing_disable(self):
        from . import qwen_train
        self.gradient_checkpointing = False
        qwen_train.set_activation_recompute(False)

    @property
    def is_gradient_checkpointing(self):
        return bool(self.gradient_checkpointing)

    def enable_input_require_grads(self):
        self._input_require_grads = True

    def resize_token_embeddings(self, new_num_tokens: int):
        old = self.model.embed_tokens.weight
        if new_num_tokens == old.shape[0]:
            return self.model.embed_tokens
        def grow(w):
            n = torch.empty((new_num_tokens, w.shape[1]), dtype=w.dtype, device=w.device)
            k = min(new_num_tokens, w.shape[0])
            n[:k] = w.data[:k]
            if new_num_tokens > k:  # HF initialises new rows from N(mean, cov-ish); mean is the deterministic part
                n[k:] = w.data.float().mean(0, keepdim=True).to(w.dtype)
            return nn.Parameter(n, requires_grad=w.requires_grad)
        self.model.embed_tokens.weight = grow(self.model.embed_tokens.weight)
        self.model.embed_tokens.num_embeddings = new_num_tokens
        self.lm_head.weight = self.model.embed_tokens.weight if self.config.tie_word_embeddings else grow(self.lm_head.weight)
        self.lm_head.out_features = new_num_tokens
        self.config.vocab_size = new_num_tokens
        return self.model.embed_tokens

    # -- multimodal embedding assembly ---------------------------------------------------------------------
    def _embed(self, pl, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw):
        """embed_tokens rows + vision features written over the placeholder rows (HF masked_scatter, modeling_qwen2_5_vl.py:1206-1223).  The index tensors
        live in the host plan `pl` (built once per distinct input: every host -> device copy of an index array is a stream sync)."""
        w = self.model.embed_tokens.weight
        c = self.config
        ids_packed_np, ids_packed_dev = pl["ids_packed_np"], pl["ids_packed"]
        if torch.is_grad_enabled() and w.requires_grad:
            from .qwen_train import EmbedFn
            if "text_rows" not in pl:
                pl["text_rows"] = np.flatnonzero((ids_packed_np != c.image_token_id) & (ids_packed_np != c.video_token_id))
            x = EmbedFn.apply(w, ids_packed_dev, ids_packed_np, pl["text_rows"], pl)
        else:
            x = ops.gather_rows(w, ids_packed_dev)
        for px, grid, tok in ((pixel_values, image_grid_thw, c.image_token_id), (pixel_values_videos, video_grid_thw, c.video_token_id)):
            if px is None:
                continue
            # a frozen tower with constant pixels needs no graph: under no_grad the windowed blocks take the attention kernel that rotates q / k while loading
            # (no stand-alone rope pass) -- the reference freezes the vision tower during RGA3 training (train_joint.py:193-251 trains LoRA, heads, mask decoder)
            frozen = not px.requires_grad and not any(p_.requires_grad for p_ in self.visual.parameters())
            emb = self._prefetched_vision(px, grid, "image" if tok == c.image_token_id else "video") if frozen else None
            if emb is None:
                with torch.no_grad() if frozen else contextlib.nullcontext():
                    emb = self.visual(px, _np(grid))
            key = ("where", tok)
            if key not in pl:
                where = np.flatnonzero(ids_packed_np == tok)
                pl[key] = (where.size, upload(where, x.device))
            n_where, where_dev = pl[key]
            if n_where != emb.shape[0]:
                raise ValueError(f"vision features and placeholder tokens do not match: tokens {n_where}, features {emb.shape[0]}")
            ops.scatter_rows_(x, where_dev, emb)
        return x

    # -- frozen vision tower, one step ahead ------------------------------------------------------------------------------
    def prefetch_vision(self, pixel_values=None, image_grid_thw=None, pixel_values_videos=None, video_grid_thw=None, **_):
        """Run the FROZEN vision tower on the NEXT batch's pixels now, on a side stream, and keep the result for the forward that receives these same tensors.
        The tower does not depend on anything the optimizer updates (the reference freezes it, train_joint.py:190-191), so a trainer that already holds the next
        batch (DataLoader prefetch) can enqueue it behind the current forward: it then fills the CUs that the launch-bound mask-path backward and the HBM-bound
        optimizer leave idle.  Optional: a forward without a matching prefetch computes the features itself; results are bit-identical either way (same kernels)."""
        if any(p_.requires_grad for p_ in self.visual.parameters()):
            return
        st = self.__dict__.get("_pf_stream")
        if st is None:
            st = self.__dict__["_pf_stream"] = torch.cuda.Stream(device=self.device)
        cache = self.__dict__.setdefault("_pf_cache", {})
        cache.clear()
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)                     # the pixels (and everything queued so far) come first
        for slot, px, grid in (("image", pixel_values, image_grid_thw), ("video", pixel_values_videos, video_grid_thw)):
            if px is None or px.requires_grad:
                continue
            grid_np = _np(grid)
            with torch.cuda.stream(st), torch.no_grad():
                st.wait_event(ready)
                px.record_stream(st)          # read on the side stream: the allocator must not recycle it before that stream is done with it
                emb = self.visual(px, grid_np)
                done = torch.cuda.Event()
                done.record(st)
            # keyed by slot + address; valid only for the same tensor version / shape / dtype AND the same grid (the features of one pixel tensor differ
            # by grid: window index, positions).  The entry holds px, so its address cannot be recycled while the entry lives.
            sig = (px._version, tuple(px.shape), px.dtype, tuple(int(v) for v in np.asarray(grid_np).reshape(-1)))
            cache[(slot, px.data_ptr())] = (sig, px, emb, done)

    def _prefetched_vision(self, px, grid, slot="video"):
        cache = self.__dict__.get("_pf_cache")
        hit = cache.pop((slot, px.data_ptr()), None) if cache else None
        if hit is None or hit[0] != (px._version, tuple(px.shape), px.dtype, tuple(int(v) for v in np.asarray(_np(grid)).reshape(-1))):
            return None
        _, _, emb, done = hit
        cur = torch.cuda.current_stream(px.device)
        cur.wait_event(done)
        emb.record_stream(cur)                # allocated on the side stream, consumed (and later freed) on this one
        return emb

    def reuse_host_plan(self, on: bool = True):
        """Opt in to keeping the host plan of the last forward for as long as the SAME unmodified tensor objects come back (an evaluation loop over a fixed
        clip, a benchmark that repeats one batch).  Off by default: a training loop feeds a fresh batch per step, and a caller that refills persistent buffers
        through raw pointers / `.data` / numpy views bypasses the version counters the reuse test relies on."""
        self.__dict__["_plan_reuse"] = bool(on)
        if not on:
            self.__dict__.pop("_plan_cache", None)
        return self

    def _host_plan(self, input_ids, attention_mask, position_ids, labels, image_grid_thw, video_grid_thw, second_per_grid_ts, past_len, dev):
        """Everything the forward derives ON THE HOST from the integer inputs (token ids, masks, labels, grids): mRoPE position ids, the packing of valid
        tokens, cu_seqlens, and their device copies.  Reading device tensors back is a device -> host sync that stops the host running ahead of the
        GPU; the plan of the last call is therefore kept and reused when the SAME tensor objects come back unmodified (same Python objects, same
        autograd version counters: an evaluation loop over a fixed clip, gradient-accumulation micro-steps, benchmark steps).  New tensors -> new plan."""
        c = self.config
        tens = (input_ids, attention_mask, position_ids, labels, image_grid_thw, video_grid_thw, second_per_grid_ts)
        sig = tuple((id(t), t._version, tuple(t.shape)) if isinstance(t, torch.Tensor) else (None if t is None else repr(t)) for t in tens) + (past_len, str(dev), c.mrope_temporal_rule)
        hit = self.__dict__.get("_plan_cache") if self.__dict__.get("_plan_reuse") else None   # opt-in (reuse_host_plan): writes that bypass the version counter would go unseen
        if hit is not None and hit[0] == sig and all((r() is t) if r is not None else True for r, t in zip(hit[1], tens)):
            return hit[2]
        ids_np = host_of(input_ids)     # the collate function's own CPU copy when dict_to_cuda attached it: no device -> host read
        B, S = ids_np.shape
        if attention_mask is not None:
            am_np = host_of(attention_mask).astype(bool)
        else:
            am_np = np.ones((B, S + past_len), dtype=bool)
        am_cur = am_np[:, -S:]
        rope_deltas = None
        # ---- 3-axis positions
        if position_ids is None:
            if past_len == 0:
                pos_np, deltas = QI.rope_index(ids_np, c.image_token_id, c.video_token_id, c.vision_config.spatial_merge_size,
                                               c.vision_config.tokens_per_second, _np(image_grid_thw), _np(video_grid_thw),
                                               _np(second_per_grid_ts), am_cur if attention_mask is not None else None,
                                               c.mrope_temporal_rule)
                rope_deltas = upload(deltas, dev)
            else:  # decode step: 1-D positions shifted by the prefill's rope delta (modeling_qwen2_5_vl.py:1160-1172)
                base = am_np.cumsum(-1)[:, -S:] - 1
                d = self.rope_deltas.cpu().numpy() if self.rope_deltas is not None else np.zeros((B, 1), dtype=np.int64)
                pos_np = np.broadcast_to((base + d)[None], (3, B, S)).copy()
        else:
            pos_np = host_of(position_ids)
            if pos_np.ndim == 2:
                pos_np = np.broadcast_to(pos_np[None], (3,) + pos_np.shape).copy()
        # ---- pack valid tokens
        lens = am_cur.sum(1)
        flat_keep = np.flatnonzero(am_cur.reshape(-1))
        ids_packed_np = ids_np.reshape(-1)[flat_keep]
        keep_dev = upload(flat_keep, dev)
        ids_packed = input_ids.reshape(-1) if (input_ids.is_cuda and flat_keep.size == B * S) else upload(ids_packed_np, dev)
        pos3 = upload(pos_np.reshape(3, -1)[:, flat_keep], dev)
        cu = upload(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), dev)
        plan = dict(ids_np=ids_np, am_cur=am_cur, lens=lens, flat_keep=flat_keep, ids_packed_np=ids_packed_np, keep_dev=keep_dev, ids_packed=ids_packed,
                    pos3=pos3, cu=cu, rope_deltas=rope_deltas, labels_np=host_of(labels))
        if past_len == 0 and self.__dict__.get("_plan_reuse"):   # decode steps change every call: not worth keeping
            import weakref
            self.__dict__["_plan_cache"] = (sig, tuple(weakref.ref(t) if isinstance(t, torch.Tensor) else None for t in tens), plan)
        return plan

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, pixel_values=None,
                pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None, rope_deltas=None, cache_position=None,
                second_per_grid_ts=None, **kwargs):
        c = self.config
        if inputs_embeds is not None:
            raise NotImplementedError("inputs_embeds entry is not part of the RGA3 hot path")
        dev = self.device
        cache = past_key_values if isinstance(past_key_values, KVCache) else None
        past_len = cache.get_seq_length() if cache is not None else 0
        pl = self._host_plan(input_ids, attention_mask, position_ids, labels, image_grid_thw, video_grid_thw, second_per_grid_ts, past_len, dev)
        ids_np, am_cur, lens, flat_keep, ids_packed_np = pl["ids_np"], pl["am_cur"], pl["lens"], pl["flat_keep"], pl["ids_packed_np"]
        keep_dev, ids_packed, pos3, cu = pl["keep_dev"], pl["ids_packed"], pl["pos3"], pl["cu"]
        B, S = ids_np.shape
        if pl["rope_deltas"] is not None:
            self.rope_deltas = pl["rope_deltas"]
        x = self._embed(pl, pixel_values if past_len == 0 else None, image_grid_thw, pixel_values_videos if past_len == 0 else None, video_grid_thw)
        self.__dict__["_last_plan"] = pl
        trainable = torch.is_grad_enabled() and labels is not None and cache is None and any(p.requires_grad for p in self.parameters())
        if trainable:
            from .qwen_train import lm_train_forward
            loss, hn = lm_train_forward(self, x, pos3, cu, int(lens.max()), pl["labels_np"], am_cur, flat_keep, lens, plan=pl)
            if flat_keep.size == B * S:
                full = hn
            else:
                from ..hip.autograd import ScatterRowsFn
                full = ScatterRowsFn.apply(hn, keep_dev, B * S)
            return CausalLMOutput(loss=loss, logits=None, past_key_values=None, hidden_states=(full.view(B, S, -1),), rope_deltas=self.rope_deltas)
        if cache is not None:
            cache.begin(lens.tolist())
        h, hs = self.model(x, pos3, cu, int(lens.max()), cache, collect_hidden=bool(output_hidden_states))
        if cache is not None:
            cache.commit()
        logits_p = self.lm_head(h)

        def unpack(t):
            if flat_keep.size == B * S:
                return t.view(B, S, -1)
            full = torch.zeros((B * S, t.shape[-1]), dtype=t.dtype, device=t.device)
            ops.scatter_rows_(full, keep_dev, t)
            return full.view(B, S, -1)

        loss = None
        if labels is not None:
            loss = self._shifted_ce(logits_p, pl["labels_np"], am_cur, flat_keep, lens, pl)
        return CausalLMOutput(loss=loss, logits=unpack(logits_p), past_key_values=cache,
                              hidden_states=tuple(unpack(t) for t in hs) if hs is not None else None, rope_deltas=self.rope_deltas)

    def _shifted_ce(self, logits_p, labels, am_cur, flat_keep, lens, pl=None):
        """Mean CE of token t's logits against label t+1 over labels != -100 (HF ForCausalLMLoss)."""
        pl = pl if pl is not None else {}
        if "ce_targets" not in pl:
            lab = host_of(labels)
            B, S = lab.shape
            nxt = np.full((B, S), -100, dtype=np.int64)
            nxt[:, :-1] = lab[:, 1:]
            pl["ce_targets"] = (upload(nxt.reshape(-1)[flat_keep], logits_p.device), int((nxt.reshape(-1)[flat_keep] != -100).sum()))
        tgt, n = pl["ce_targets"]
        row_loss = ops.cross_entropy_rows(logits_p, tgt)
        return row_loss.sum() / max(n, 1)

    # -- generation (greedy / sampling) --------------------------------------------------------------------
    @torch.no_grad()
    def _capture_decode_step(self, cache, tok, pos_host):
        """First decode step of a single-sequence generate(): run it eagerly on a side stream (tuner picks, workspaces), then capture the same
        step over static inputs (token id, position, device-side cache length).  Returns the replay state; its "logits" are those of the step
        just taken.  The eager step and the captured one write the same cache row, so the cache is consistent whichever ran last."""
        dev = tok.device
        n_stored = cache.lens[0]
        st = {"tok": tok.clone(), "pos3": torch.from_numpy(np.broadcast_to(pos_host[None], (3, 1)).copy()).to(dev), "kv": StaticKVCache(cache, dev)}
        st["cu_q"] = torch.tensor([0, 1], dtype=torch.int32, device=dev)   # every tensor the captured launches read must outlive the graph: keep it in st

        def body():
            h, _ = self.model(ops.gather_rows(self.model.embed_tokens.weight, st["tok"]), st["pos3"], st["cu_q"], 1, st["kv"])
            lg = self.lm_head(h)
            st["pos3"] += 1          # state of the NEXT step, advanced inside the graph
            st["kv"].advance()
            return lg

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with ops.workspace_scope(("decode-graph", id(st))):     # the scratch the capture bakes in belongs to THIS graph, not to a stream handle (ops.workspace_scope)
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream().wait_stream(side)
            st["pos3"] -= 1                # rewind the state the warm-up advanced; the warm-up wrote cache row n_stored, the capture rewrites it
            st["kv"].reset(n_stored)
            st["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st["graph"]):
                st["logits"] = body()
        st["graph"].replay()          # capturing executes nothing: the state still says "this step" -- now run it for real
        cache.lens[0] = n_stored      # the host-side bookkeeping of the eager cache is not used any more during this call
        return st

    def generate(self, input_ids=None, attention_mask=None, max_new_tokens=128, do_sample=False, temperature=1.0, top_p=1.0,
                 eos_token_id=None, pixel_values=None, pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None,
                 second_per_grid_ts=None, **kwargs):
        """Prefill + KV-cached decode; returns [B, S + new] token ids (reference app.py:308-317 usage)."""
        c = self.config
        B, S = input_ids.shape
        dev = self.device
        eos = eos_token_id if eos_token_id is not None else c.eos_token_id
        eos = set(eos if isinstance(eos, (list, tuple)) else [eos])
        # single sequences decode by replaying a captured step: its KV buffers and graph are kept on the model per capacity bucket, so only the
        # first generate() of a given length class pays for the capture
        graph_ok = (B == 1 and max_new_tokens >= 16 and not torch.is_grad_enabled() and kwargs.get("decode_graph", True)
                    and all(_is_plain(l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj) for l in self.model.layers))
        dstate = None
        if graph_ok:
            cap = (S + max_new_tokens + 1023) // 1024 * 1024
            dstate = self.__dict__.setdefault("_decode_states", {}).setdefault((cap, str(dev), self.dtype), {})
            # the captured launches read the weights (and the packs built from them) through their pointers: any in-place update since the
            # capture bumps a parameter version and the graph is dropped (re-captured on this call)
            sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
            if dstate.get("sig") != sig:
                dstate.clear()
                dstate["sig"] = sig
            if "cache" not in dstate:
                dstate["cache"] = KVCache(c.num_hidden_layers, 1, cap, c.num_key_value_heads, c.head_dim, dev, self.dtype)
            cache = dstate["cache"]
            cache.lens = [0]
        else:
            cache = KVCache(c.num_hidden_layers, B, S + max_new_tokens, c.num_key_value_heads, c.head_dim, dev, self.dtype)
        am = attention_mask if attention_mask is not None else torch.ones_like(input_ids)
        out = self.forward(input_ids=input_ids, attention_mask=am, past_key_values=cache, pixel_values=pixel_values,
                           pixel_values_videos=pixel_values_videos, image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw,
                           second_per_grid_ts=second_per_grid_ts)
        last_idx = (am.long().cumsum(-1).argmax(-1))  # last valid position per row
        logits = out.logits[torch.arange(B, device=dev), last_idx.to(dev)]
        seqs = input_ids
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        pad = c.pad_token_id if c.pad_token_id is not None else next(iter(eos))
        eos_t = torch.tensor(sorted(eos), device=dev)
        new_cols, all_done = [], []
        deltas = self.rope_deltas.cpu().numpy().reshape(B) if self.rope_deltas is not None else np.zeros(B, dtype=np.int64)
        pos_host = am.detach().cpu().numpy().astype(bool).sum(1).astype(np.int64) - 1 + deltas    # mrope position of the last prefill token (modeling_qwen2_5_vl.py:1160-1172)
        cu_dec = torch.arange(B + 1, dtype=torch.int32, device=dev)
        dec = None
        # The "everyone has emitted EOS" test is a device -> host sync; taken every step it keeps the host from running ahead of the GPU and
        # exposes every launch.  It is taken every SYNC_EVERY steps instead: a finished batch may run up to SYNC_EVERY - 1 extra steps, whose
        # columns (all pad by construction) are cut below, so the returned ids are exactly those of the step-by-step loop.
        SYNC_EVERY = 8
        for step in range(max_new_tokens):
            if do_sample:
                pr = torch.softmax(logits.float() / max(temperature, 1e-5), -1)
                nxt = torch.multinomial(pr, 1)[:, 0]
            else:
                nxt = logits.float().argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
            new_cols.append(nxt[:, None])
            am = torch.cat([am, (~done).to(am.dtype)[:, None]], dim=1)
            done = done | torch.isin(nxt, eos_t)
            all_done.append(done.all())
            if step + 1 == max_new_tokens or ((step + 1) % SYNC_EVERY == 0 and bool(all_done[-1])):
                break
            # decode step without any device -> host traffic: the host knows every position (valid prefill tokens + steps + the prefill's
            # rope delta) and the token ids stay on the device.  Finished sequences keep stepping on pad tokens until the next check (their
            # outputs are forced to pad above and nothing of theirs is read by the others), which is what keeps the step free of syncs.
            pos_host += 1
            if graph_ok:
                # one sequence: the whole step (embedding row, 28 layers, final norm, LM head: ~400 launches of a few microseconds, bound by
                # launch latency) is captured once (kept on the model per capacity bucket) after a first eager step and replayed per token
                if dec is None and "graph" in dstate:          # captured by an earlier call: point the static state at this sequence
                    dec = dstate
                    dec["pos3"].copy_(torch.from_numpy(np.broadcast_to(pos_host[None], (3, 1)).copy()).to(dev))
                    dec["kv"].reset(cache.lens[0])
                    dec["tok"].copy_(nxt)
                    dec["graph"].replay()
                elif dec is None:
                    dec = self._capture_decode_step(cache, nxt, pos_host)
                    dstate.update(dec)
                    dec = dstate
                else:
                    dec["tok"].copy_(nxt)
                    dec["graph"].replay()
                logits = dec["logits"]
                continue
            pos3 = torch.from_numpy(np.broadcast_to(pos_host[None], (3, B)).copy()).to(dev)
            cache.begin([1] * B)
            h, _ = self.model(ops.gather_rows(self.model.embed_tokens.weight, nxt), pos3, cu_dec, 1, cache)
            cache.commit()
            logits = self.lm_head(h)
        flags = torch.stack(all_done).cpu().numpy() if all_done else np.zeros(0, dtype=bool)
        keep = int(np.argmax(flags)) + 1 if flags.any() else len(new_cols)   # columns up to and including the step at which the last sequence finished
        ops.poll_gemm_health()
        return torch.cat([seqs] + new_cols[:keep], dim=1)


def _np(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return host_of(x)     # a device tensor moved by dict_to_cuda still has its CPU companion: no read-back
    return np.asarray(x)
