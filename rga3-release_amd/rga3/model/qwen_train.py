"""Training-step autograd for the Qwen2.5 decoder on HIP kernels (LoRA fine-tuning as in reference train_joint.py).

What is trainable on the LLM side of the reference recipe (train_joint.py:193-251, run_torchrun.sh:30-31): LoRA r=128 / alpha=256
on q_proj and v_proj of every decoder layer, embed_tokens, lm_head (plus text_hidden_fcs and the SAM2 mask decoder on the mask
path).  Base projections, norms and the whole ViT are frozen.

Design: ONE autograd node per decoder layer (DecoderLayerFn).  Its forward runs the fused inference kernels under no_grad and
keeps only the layer input (activation checkpointing == reference's gradient_checkpointing_enable, train_joint.py:188); its
backward recomputes the layer and back-propagates by hand with the same kernel library: dX GEMMs against cached transposed
packs of the frozen weights, flash-style attention backward, fused RMSNorm / SwiGLU backward, inverse RoPE, and the small
LoRA dA/dB products.  The LM head only evaluates the rows that carry a label.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn as nn

from ..hip import ops
from ..utils.staging import upload
from . import qwen2_5_vl as _Q
from .qwen2_5_vl import GatedMLP, Linear, _versions


# ------------------------------------------------------------------------------------------------ LoRA
class LoRALinear(Linear):
    """nn.Linear + low-rank update, parameter names as PEFT writes them (<name>.lora_A.default.weight, ...)."""

    def __init__(self, base: Linear, r: int, alpha: float, dropout: float = 0.0):
        super().__init__(base.in_features, base.out_features, bias=base.bias is not None, device=base.weight.device, dtype=base.weight.dtype)
        self.weight, self.bias = base.weight, base.bias
        self.weight.requires_grad_(False)
        if self.bias is not None:
            self.bias.requires_grad_(False)
        mk = lambda i, o: Linear(i, o, bias=False, device=base.weight.device, dtype=base.weight.dtype)
        self.lora_A = nn.ModuleDict({"default": mk(base.in_features, r)})
        self.lora_B = nn.ModuleDict({"default": mk(r, base.out_features)})
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B["default"].weight)
        self.scaling = alpha / r
        self.r = r
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"lora dropout {dropout}")
        self.dropout_p = float(dropout)   # nn.Dropout on the input of lora_A only, training mode only (PEFT LoraLayer)

    def forward(self, x, residual=None, act="none"):
        y = super().forward(x, residual=residual, act=act)
        xin = x.reshape(-1, x.shape[-1])
        if self.training and self.dropout_p > 0.0:
            xin = ops.dropout(xin.contiguous(), self.dropout_p, next_dropout_seed(0, 3))
        t = self.lora_A["default"](xin)
        bs = (self.lora_B["default"].weight.detach() * self.scaling).to(t.dtype)
        y2 = y.view(-1, y.shape[-1])
        ops.gemm(t, bs, residual=y2, out=y2)
        return y


def add_lora(model: nn.Module, r=128, alpha=256, dropout=0.0, targets=("q_proj", "v_proj"),
             exclude=("sam_model", "grounding_encodervisual", "text_hidden_fcs")):
    """Replace matching Linear modules by LoRALinear, with the name filter of reference train_joint.py:199-212 (including its
    missing comma, which makes SAM2's q_proj / v_proj LoRA targets too)."""
    hits = []
    for name, mod in list(model.named_modules()):
        if isinstance(mod, nn.Linear) and not isinstance(mod, LoRALinear) and all(x not in name for x in exclude) and any(x in name for x in targets):
            parent = model
            parts = name.split(".")
            for p in parts[:-1]:
                parent = getattr(parent, p)
            setattr(parent, parts[-1], LoRALinear(mod, r, alpha, dropout))
            hits.append(name)
    return hits


# ------------------------------------------------------------------------------------------------ transposed weight packs
def _wt(owner, key, *srcs, build):
    """Cached derived tensor keyed by the versions of its sources (frozen weights never change -> built once)."""
    cache = owner.__dict__.setdefault("_wt_cache", {})
    ver = _versions(*srcs)
    hit = cache.get(key)
    if hit is None or hit[0] != ver:
        with torch.no_grad():
            cache[key] = (ver, build())
    return cache[key][1]


# ------------------------------------------------------------------------------------------------ fp8 frozen-weight GEMMs (config 5)
_fp8 = {"on": False}


def set_fp8_frozen_gemms(on: bool, vision: bool = True):
    """Run the frozen-weight contractions -- the decoder's qkv / o / gate-up / down and their transposes in backward, and (vision=True) the frozen vision tower's
    qkv / proj / gate-up / down -- through the e4m3 GEMM (BASELINE.json configs[4]).  Activations are quantised per token on the fly, weights per output row once
    (cached per weight version); LoRA factors, lm_head, embeddings, norms, attention, the patch embedding and the merger stay bf16."""
    _fp8["on"] = bool(on)
    _Q._VIT_BLOCK_FP8[0] = vision_block_forward_fp8 if (on and vision) else None


def fp8_frozen_gemms() -> bool:
    return _fp8["on"]


def _fp8_pack(owner, key, *srcs, build):
    """(q uint8 [N, K], scale f32 [N]) of the weight matrix `build()` returns, cached like _wt (the bf16 matrix is not kept)."""
    def mk():
        w = build()
        q, sc = ops.quant_fp8_rows(w.contiguous())
        return (q, sc)
    return _wt(owner, "fp8:" + key, *srcs, build=mk)


_fp8_tap = [None]     # tests only: callable(key, codes uint8 [M, K], scales f32 [M]) sees the e4m3 operand of every frozen contraction (tests/test_fulldepth_parity_gpu.py
                      # hands the SAME codes to the oracle's e4m3 step, so that the comparison prices the kernels, not code flips of one-ulp activation differences)


def _fgemm(x, owner, key, srcs, build, bias=None, residual=None, out=None):
    """x [M, K] @ build()^T for a FROZEN weight: e4m3 when enabled and K % 128 == 0, else the bf16 GEMM on the cached bf16 matrix."""
    if _fp8["on"] and x.shape[1] % 128 == 0:
        q, sc = _fp8_pack(owner, key, *srcs, build=build)
        xq, xs = ops.quant_fp8_rows(x)
        if _fp8_tap[0] is not None:
            _fp8_tap[0](key, xq, xs)
        return ops.gemm_fp8(xq, xs, q, sc, bias=bias, residual=residual, out=out)
    w = _wt(owner, key, *srcs, build=build)
    return ops.gemm(x, w, bias, residual=residual, out=out)


def _fgemm_q(xq, xs, owner, key, srcs, build, bias=None, residual=None, out=None):
    """_fgemm for an activation that is ALREADY quantised (the producer kernel emitted e4m3 + row scales: ops.swiglu_fwd_quant / swiglu_bwd_quant)."""
    q, sc = _fp8_pack(owner, key, *srcs, build=build)
    if _fp8_tap[0] is not None:
        _fp8_tap[0](key, xq, xs)
    return ops.gemm_fp8(xq, xs, q, sc, bias=bias, residual=residual, out=out)


def _fp8_fused(k: int) -> bool:
    return _fp8["on"] and k % 128 == 0


def vision_block_forward_fp8(blk, x, cu, max_len, cos, sin):
    """One frozen Qwen2.5-VL vision block (HF modeling_qwen2_5_vl.py:290-321: x + attn(norm1(x)), x + mlp(norm2(x))) with its four contractions in e4m3: the
    configs[4] fine-tune runs the tower on 32 frames per micro-step (21 TFLOP, a fifth of the step's GEMM time in bf16).  Same operation order as the bf16 route;
    the SwiGLU kernel emits the e4m3 operand of the down projection directly.  Inference only (the tower is frozen: reference train_joint.py:190-191)."""
    at, mlp = blk.attn, blk.mlp
    N = x.shape[0]
    H, D = at.num_heads, at.head_dim
    h1 = blk.norm1(x)
    qkv = _fgemm(h1, at, "wqkv", (at.qkv.weight,), lambda: at.qkv.weight.detach(), bias=at.qkv.bias).view(N, 3 * H, D)
    if ops.attn_rope_win_ok(max_len, D):
        att = ops.attn_varlen_rope(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], cu, cu, max_len, D ** -0.5, cos, sin, causal=False, rope_k=True)
    else:
        ops.rope_(qkv, cos, sin, 0, 2 * H)
        att = ops.attn_varlen(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], cu, cu, max_len, D ** -0.5, causal=False, max_k=max_len)
    x1 = _fgemm(att.view(N, H * D), at, "wproj", (at.proj.weight,), lambda: at.proj.weight.detach(), bias=at.proj.bias, residual=x)
    h2 = blk.norm2(x1)
    wgu, bgu, wd = mlp._packed()
    gu = _fgemm(h2, mlp, "wgu", (mlp.gate_proj.weight, mlp.up_proj.weight), lambda: wgu, bias=bgu)
    return _down_from_gu(mlp, gu, wd, x1)


def _lora_parts(lin):
    if isinstance(lin, LoRALinear):
        return lin.lora_A["default"].weight, lin.lora_B["default"].weight, lin.scaling
    return None


# ---- derived LoRA operands, rebuilt once per optimizer step for ALL modules with a handful of batched launches.
# y = W x + s B A drop(x) is evaluated as  t_s = drop(x) (sA)^T,  y += t_s B^T;  backward  dB = dy^T t_s,  dt_s = dy (sB),  dA = dt_s^T drop(x),
# dx += dt_s A  -- the scaling s sits in the SMALL operand of exactly one product per gradient, so no activation-sized tensor is ever rescaled, and the
# two NN-form products read A^T and (sB)^T.  Round 2 built those per layer and per step with ~14 tiny launches per layer (mul, transpose, float / mul /
# cast of dB): ~400 launches of a few microseconds in a step.  Here: stack -> one mul / one transposed copy per module family.
_lora_cache = {}   # id(LoRALinear) -> (key of its parameters, sA [r, in], A^T [in, r], (sB)^T [r, out])


def _lora_versions(lin):
    A, B = lin.lora_A["default"].weight, lin.lora_B["default"].weight
    return A._version, A.data_ptr(), B._version, B.data_ptr()


def _lora_build(mods):
    groups = {}
    for m in mods:
        A, B = m.lora_A["default"].weight, m.lora_B["default"].weight
        groups.setdefault((tuple(A.shape), tuple(B.shape), float(m.scaling), A.dtype, A.device), []).append(m)
    with torch.no_grad():
        for (_, _, sc, _, _), grp in groups.items():
            A = torch.stack([m.lora_A["default"].weight.detach() for m in grp])    # [n, r, in]
            B = torch.stack([m.lora_B["default"].weight.detach() for m in grp])    # [n, out, r]
            As = A * sc
            At = A.transpose(1, 2).contiguous()
            Bts = (B * sc).transpose(1, 2).contiguous()
            for i, m in enumerate(grp):
                _lora_cache[id(m)] = (_lora_versions(m), As[i], At[i], Bts[i])


def lora_refresh(layers):
    """Call once per forward: rebuilds the derived operands of every LoRA projection of `layers` if the optimizer has stepped since they were made."""
    mods = [p for layer in layers for p in (layer.self_attn.q_proj, layer.self_attn.v_proj) if isinstance(p, LoRALinear)]
    if mods and any((e := _lora_cache.get(id(m))) is None or e[0] != _lora_versions(m) for m in (mods[0], mods[-1])):
        _lora_build(mods)
    ats = [layer.self_attn for layer in layers if _lora_cat_ok(layer.self_attn)]
    if ats and len({(a.num_heads, a.num_kv, a.head_dim, a.q_proj.lora_A["default"].weight.shape[0], float(a.q_proj.scaling), float(a.v_proj.scaling)) for a in ats}) == 1:
        if any((e := _lora_cat.get(id(a))) is None or e[0] != _lora_versions(a.q_proj) + _lora_versions(a.v_proj) for a in (ats[0], ats[-1])):
            _lora_cat_build(ats)


# ---- LoRA's B-side products folded into the frozen q|k|v product and its transpose (rga3_gemm_cat_bf16):
#   forward   qkv = h W^T + [t_q | t_v] W2^T,   W2 [(Hq + 2 Hk) D, 2 r] = [B_q 0; 0 0; 0 B_v]            (two skinny launches per layer less, one rounding less)
#   backward  [dh | dt_q | dt_v] = dqkv [W | Wn^T],   Wn [2 r, (Hq + 2 Hk) D] = [(s B_q)^T 0 0; 0 0 (s B_v)^T]   (two more)
# The block matrices of ALL layers are refreshed with the other derived operands, once per optimizer step (zeros stay; only the B blocks are rewritten).
_LORA_FOLD = os.environ.get("RGA3_LORA_FOLD", "1") != "0"
_SWIGLU_PRE = os.environ.get("RGA3_SWIGLU_PRE", "1") != "0"     # training forward: SwiGLU inside the gate | up product, pre-activations stored by the same launch
_lora_cat = {}     # id(attention module) -> (key, W2, Wn)
_lora_cat_store = {}   # (ids of the module set, Nqkv, r, dtype, device) -> (W2_all, Wn_all)


def set_lora_fold(on: bool):
    """A/B switch (tests, bench): fold LoRA's B-side products into the frozen q|k|v products."""
    global _LORA_FOLD
    _LORA_FOLD = bool(on)


def _lora_cat_ok(at) -> bool:
    q, v = at.q_proj, at.v_proj
    if not (_LORA_FOLD and isinstance(q, LoRALinear) and isinstance(v, LoRALinear)) or isinstance(at.k_proj, LoRALinear):
        return False
    rq, rv = q.lora_A["default"].weight.shape[0], v.lora_A["default"].weight.shape[0]
    return rq == rv and (2 * rq) % 64 == 0 and q.lora_A["default"].weight.dtype == torch.bfloat16 and q.weight.shape[1] % 64 == 0


def _lora_cat_build(ats):
    with torch.no_grad():
        at0 = ats[0]
        Hq, Hk, D = at0.num_heads, at0.num_kv, at0.head_dim
        r = at0.q_proj.lora_A["default"].weight.shape[0]
        Nq, Nall = Hq * D, (Hq + 2 * Hk) * D
        Bq = torch.stack([a.q_proj.lora_B["default"].weight.detach() for a in ats])          # [L, Hq D, r]
        Bv = torch.stack([a.v_proj.lora_B["default"].weight.detach() for a in ats])          # [L, Hk D, r]
        # keyed by the IDENTITY of the module set: keyed by shape alone, the per-module fallback of _lora_cat_ops gave every layer the same (1, ...)
        # buffer, each build overwriting the previous layer's B blocks -- the backward of all layers but the last then read the last layer's (sB)^T
        ids = tuple(id(a) for a in ats)
        key = (ids, Nall, r, Bq.dtype, Bq.device)
        st = _lora_cat_store.get(key)
        if st is None:
            for k in [k for k in _lora_cat_store if k != key and not set(k[0]).isdisjoint(ids)]:    # a module belongs to ONE live set: drop the sets it leaves
                del _lora_cat_store[k]
            st = (torch.zeros((len(ats), Nall, 2 * r), dtype=Bq.dtype, device=Bq.device), torch.zeros((len(ats), 2 * r, Nall), dtype=Bq.dtype, device=Bq.device))
            _lora_cat_store[key] = st
        W2, Wn = st
        W2[:, :Nq, :r] = Bq
        W2[:, Nq + Hk * D:, r:] = Bv
        Wn[:, :r, :Nq] = (Bq * float(at0.q_proj.scaling)).transpose(1, 2)
        Wn[:, r:, Nq + Hk * D:] = (Bv * float(at0.v_proj.scaling)).transpose(1, 2)
        for i, a in enumerate(ats):
            _lora_cat[id(a)] = (_lora_versions(a.q_proj) + _lora_versions(a.v_proj), W2[i], Wn[i])


def _lora_cat_ops(at):
    """(W2, Wn) of one attention module, current with its LoRA parameters."""
    e = _lora_cat.get(id(at))
    if e is None or e[0] != _lora_versions(at.q_proj) + _lora_versions(at.v_proj):
        _lora_cat_build([at])
        e = _lora_cat[id(at)]
    return e[1], e[2]


def _lora_ops(lin):
    """(sA, A^T, (sB)^T) of one projection, current with its parameters (a stale or missing entry is rebuilt for this module alone)."""
    e = _lora_cache.get(id(lin))
    if e is None or e[0] != _lora_versions(lin):
        _lora_build([lin])
        e = _lora_cache[id(lin)]
    return e[1], e[2], e[3]


_drop_state = {"step": 0}


def next_dropout_seed(layer_idx: int, which: int, advance: bool = True) -> int:
    """63-bit seed of one dropout mask: (torch's global seed, forward-pass counter, layer, projection).  The mask itself is a counter
    hash of (seed, element index) inside the kernel, so storing the seed is storing the mask."""
    if advance:
        _drop_state["step"] += 1
    base = torch.initial_seed() & 0xFFFFFF
    return ((base << 39) ^ (_drop_state["step"] << 12) ^ (layer_idx << 2) ^ which) & 0x7FFFFFFFFFFFFFFF


def preview_dropout_seeds(n_layers: int):
    """The (seed_q, seed_v) pairs the NEXT lm_train_forward will use for layers 0..n_layers-1 (tests reproduce the masks from them)."""
    saved = _drop_state["step"]
    try:
        return [(next_dropout_seed(li, 0), next_dropout_seed(li, 1, advance=False)) for li in range(n_layers)]
    finally:
        _drop_state["step"] = saved


def _lora_dropout(at):
    """(p_q, p_v) active dropout probabilities of the two LoRA branches (0 outside training mode)."""
    pq = at.q_proj.dropout_p if isinstance(at.q_proj, LoRALinear) and at.q_proj.training else 0.0
    pv = at.v_proj.dropout_p if isinstance(at.v_proj, LoRALinear) and at.v_proj.training else 0.0
    return pq, pv


def _drop_pair(h1, pq, pv, seeds):
    """(dropout_q(h1), dropout_v(h1)): both masks in one launch that reads h1 once when both branches drop (ops.dropout_pair; same masks as ops.dropout)."""
    if pq > 0.0 and pv > 0.0 and h1.is_contiguous():
        return ops.dropout_pair(h1, pq, seeds[0], h1, pv, seeds[1])
    return (ops.dropout(h1, pq, seeds[0]) if pq > 0.0 else h1), (ops.dropout(h1, pv, seeds[1]) if pv > 0.0 else h1)


def qkv_with_lora(at, h1, seeds=None, want_inputs=False):
    """Fused q/k/v projection (+ LoRA updates on q and v accumulated in place); returns (qkv2 [T, (Hq+2Hk)D], tq, tv[, hq, hv]).
    seeds = (seed_q, seed_v) of the LoRA-branch dropout masks (None: fresh seeds when dropout is active); hq / hv are the (dropped)
    inputs of lora_A, which the backward needs for dA."""
    Hq, Hk, D = at.num_heads, at.num_kv, at.head_dim
    wqkv, bqkv = at._packed()
    lq, lv = _lora_parts(at.q_proj), _lora_parts(at.v_proj)
    pq, pv = _lora_dropout(at)
    if seeds is None and (pq > 0.0 or pv > 0.0):
        lid = getattr(at, "layer_idx", 0) or 0
        seeds = (next_dropout_seed(lid, 0), next_dropout_seed(lid, 1, advance=False))
    if not _fp8["on"] and h1.shape[0] > 16 and _lora_cat_ok(at):
        # B side folded into the frozen product: t_q, t_v first (column halves of one buffer), then ONE product over K = H + 2 r
        r = lq[0].shape[0]
        hq, hv = _drop_pair(h1, pq, pv, seeds)
        tt = torch.empty((h1.shape[0], 2 * r), dtype=h1.dtype, device=h1.device)
        tq = ops.gemm(hq, _lora_ops(at.q_proj)[0], out=tt[:, :r])          # t_s = drop(h) (sA)^T
        tv = ops.gemm(hv, _lora_ops(at.v_proj)[0], out=tt[:, r:])
        qkv2 = ops.gemm_cat(h1, wqkv, bqkv, a2=tt, w2=_lora_cat_ops(at)[0])
        return (qkv2, tq, tv, hq, hv) if want_inputs else (qkv2, tq, tv)
    if _fp8["on"] and h1.shape[1] % 128 == 0:
        qkv2 = _fgemm(h1, at, "wqkv", (at.q_proj.weight, at.k_proj.weight, at.v_proj.weight), lambda: wqkv, bias=bqkv)
    else:
        qkv2 = ops.gemm(h1, wqkv, bqkv)
    tq = tv = None
    hq = hv = h1
    if lq is not None and lv is not None and pq > 0.0 and pv > 0.0:
        hq, hv = _drop_pair(h1, pq, pv, seeds)
        pq = pv = 0.0                                          # (done)
    if lq is not None:
        if pq > 0.0:
            hq = ops.dropout(h1, pq, seeds[0])
        tq = ops.gemm(hq, _lora_ops(at.q_proj)[0])            # t_s = drop(h) (sA)^T
        oq = qkv2[:, : Hq * D]
        ops.gemm(tq, lq[1].detach(), residual=oq, out=oq)       # += t_s B^T
    if lv is not None:
        if pv > 0.0:
            hv = ops.dropout(h1, pv, seeds[1])
        tv = ops.gemm(hv, _lora_ops(at.v_proj)[0])
        ov = qkv2[:, (Hq + Hk) * D:]
        ops.gemm(tv, lv[1].detach(), residual=ov, out=ov)
    return (qkv2, tq, tv, hq, hv) if want_inputs else (qkv2, tq, tv)


def _layer_forward_fp8(layer, x, cos, sin, cu, max_len, seeds):
    """Decoder layer forward with the frozen contractions in e4m3 (same operation order as the recompute in DecoderLayerFn.backward)."""
    at, mlp = layer.self_attn, layer.mlp
    Hq, Hk, D = at.num_heads, at.num_kv, at.head_dim
    T = x.shape[0]
    w1, w2 = layer.input_layernorm, layer.post_attention_layernorm
    h1 = ops.rmsnorm(x, w1.weight, w1.variance_epsilon)
    qkv2 = qkv_with_lora(at, h1, seeds=seeds)[0]
    qkv = qkv2.view(T, Hq + 2 * Hk, D)
    ops.rope_(qkv, cos, sin, 0, Hq + Hk)
    att = ops.attn_varlen(qkv[:, :Hq], qkv[:, Hq:Hq + Hk], qkv[:, Hq + Hk:], cu, cu, max_len, D ** -0.5, causal=True)
    x1 = _fgemm(att.view(T, Hq * D), at, "wo", (at.o_proj.weight,), lambda: at.o_proj.weight.detach(), residual=x)
    h2 = ops.rmsnorm(x1, w2.weight, w2.variance_epsilon)
    wgu, bgu, wd = mlp._packed()
    gu = _fgemm(h2, mlp, "wgu", (mlp.gate_proj.weight, mlp.up_proj.weight), lambda: wgu, bias=bgu)
    return _down_from_gu(mlp, gu, wd, x1)


def _down_from_gu(mlp, gu, wd, x1):
    """down(silu(gate) * up) + x1 from the interleaved pre-activations.  Under the fp8 switch the SwiGLU kernel emits the e4m3 operand of the down
    projection directly (its bf16 form is never written: 2 x 2 bytes per element less traffic on the widest activation of the layer)."""
    if _fp8_fused(gu.shape[1] // 2):
        aq, asc = ops.swiglu_fwd_quant(gu)
        return _fgemm_q(aq, asc, mlp, "wd", (mlp.down_proj.weight,), lambda: wd, bias=mlp.down_proj.bias, residual=x1)
    return _fgemm(ops.swiglu_fwd(gu), mlp, "wd", (mlp.down_proj.weight,), lambda: wd, bias=mlp.down_proj.bias, residual=x1)


_acts = {"store": True}


def set_activation_recompute(on: bool):
    """True: a decoder layer saves only its input and recomputes the rest in backward (what the reference's gradient checkpointing does,
    train_joint.py gradient_checkpointing_enable).  False (default here): the layer keeps what its backward reads -- normed inputs, roped
    q/k/v, attention output + LSE, the MLP pre-activations: ~290 MB per layer at S = 2112, 8 GB for 28 layers of the 288 GB -- and
    backward skips the recompute (a qkv, an o-proj and a gate-up GEMM plus an attention forward per layer).  Same arithmetic either way."""
    _acts["store"] = not on


def _layer_forward_store(layer, x, cos, sin, cu, max_len, seeds):
    """Decoder layer forward for training that returns (y, saved): saved = exactly the tensors DecoderLayerFn.backward otherwise recomputes
    (same calls in the same order, frozen contractions through _fgemm so the fp8 switch applies)."""
    at, mlp = layer.self_attn, layer.mlp
    Hq, Hk, D = at.num_heads, at.num_kv, at.head_dim
    T = x.shape[0]
    w1, w2 = layer.input_layernorm, layer.post_attention_layernorm
    h1 = ops.rmsnorm(x, w1.weight, w1.variance_epsilon)
    qkv2, tq, tv, hq_in, hv_in = qkv_with_lora(at, h1, seeds=seeds, want_inputs=True)
    qkv = qkv2.view(T, Hq + 2 * Hk, D)
    ops.rope_(qkv, cos, sin, 0, Hq + Hk)
    att, lse = ops.attn_varlen(qkv[:, :Hq], qkv[:, Hq:Hq + Hk], qkv[:, Hq + Hk:], cu, cu, max_len, D ** -0.5, causal=True, return_lse=True)
    x1 = _fgemm(att.view(T, Hq * D), at, "wo", (at.o_proj.weight,), lambda: at.o_proj.weight.detach(), residual=x)
    h2 = ops.rmsnorm(x1, w2.weight, w2.variance_epsilon)
    wgu, bgu, wd = mlp._packed()
    if not _fp8["on"] and _SWIGLU_PRE and T > 4:
        # the gate | up product applies SwiGLU in its epilogue AND leaves the pre-activations the backward needs: no stand-alone SwiGLU launch
        a_, gu = ops.gemm_swiglu_pre(h2, wgu, bgu)
        y = _fgemm(a_, mlp, "wd", (mlp.down_proj.weight,), lambda: wd, bias=mlp.down_proj.bias, residual=x1)
    else:
        gu = _fgemm(h2, mlp, "wgu", (mlp.gate_proj.weight, mlp.up_proj.weight), lambda: wgu, bias=bgu)
        y = _down_from_gu(mlp, gu, wd, x1)
    return y, (h1, qkv, att, lse, x1, h2, gu, tq, tv, hq_in, hv_in)


_neg = {}


def _neg_table(sin):
    """-sin, made once per table (every layer's backward un-rotates with the same one)."""
    key = (sin.data_ptr(), sin._version, tuple(sin.shape))
    if _neg.get("key") != key:
        _neg["key"], _neg["t"] = key, (-sin).contiguous()
        _neg["src"] = sin          # keeps the source alive: its address cannot be reused while the entry exists
    return _neg["t"]


class DecoderLayerFn(torch.autograd.Function):
    """y = DecoderLayer(x).  Keeps the layer's intermediates for backward (default) or only x and recomputes (set_activation_recompute)."""

    @staticmethod
    def forward(ctx, x, aq, bq, av, bv, layer, cos, sin, cu, max_len, seeds):
        at0 = layer.self_attn
        ctx.stored = False
        if _acts["store"]:
            with torch.no_grad():
                y, saved = _layer_forward_store(layer, x, cos, sin, cu, max_len, seeds)
            ctx.stored = True
            ctx.seeds = seeds
            ctx.layer, ctx.cos, ctx.sin, ctx.cu, ctx.max_len = layer, cos, sin, cu, max_len
            ctx.save_for_backward(x, *saved)
            return y
        if _fp8["on"]:
            with torch.no_grad():
                y = _layer_forward_fp8(layer, x, cos, sin, cu, max_len, seeds)
        else:
            at0._lora_drop_seeds = seeds          # read by DecoderAttention's native-LoRA path (qwen2_5_vl.py) for this call only
            try:
                with torch.no_grad():
                    y = layer(x, cos, sin, cu, max_len, None)
            finally:
                at0._lora_drop_seeds = None
        ctx.seeds = seeds
        ctx.layer, ctx.cos, ctx.sin, ctx.cu, ctx.max_len = layer, cos, sin, cu, max_len
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x = ctx.saved_tensors[0]
        layer, cos, sin, cu, max_len = ctx.layer, ctx.cos, ctx.sin, ctx.cu, ctx.max_len
        at, mlp = layer.self_attn, layer.mlp
        Hq, Hk, D = at.num_heads, at.num_kv, at.head_dim
        T = x.shape[0]
        dy = dy.contiguous()
        with torch.no_grad():
            w1, w2 = layer.input_layernorm, layer.post_attention_layernorm
            wqkv, bqkv = at._packed()
            lq, lv = _lora_parts(at.q_proj), _lora_parts(at.v_proj)
            pq, pv = _lora_dropout(at)
            wgu, bgu, wd = mlp._packed()
            if ctx.stored:
                h1, qkv, att, lse, x1, h2, gu, tq, tv, hq_in, hv_in = ctx.saved_tensors[1:]
            else:
                # ---- recompute forward pieces the backward needs
                h1 = ops.rmsnorm(x, w1.weight, w1.variance_epsilon)
                qkv2, tq, tv, hq_in, hv_in = qkv_with_lora(at, h1, seeds=ctx.seeds, want_inputs=True)
                qkv = qkv2.view(T, Hq + 2 * Hk, D)
                ops.rope_(qkv, cos, sin, 0, Hq + Hk)
                att, lse = ops.attn_varlen(qkv[:, :Hq], qkv[:, Hq:Hq + Hk], qkv[:, Hq + Hk:], cu, cu, max_len, D ** -0.5, causal=True, return_lse=True)
                x1 = _fgemm(att.view(T, Hq * D), at, "wo", (at.o_proj.weight,), lambda: at.o_proj.weight.detach(), residual=x)
                h2 = ops.rmsnorm(x1, w2.weight, w2.variance_epsilon)
                gu = _fgemm(h2, mlp, "wgu", (mlp.gate_proj.weight, mlp.up_proj.weight), lambda: wgu, bias=bgu)   # pre-activations, interleaved [T, 2*Ip]
            q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hk], qkv[:, Hq + Hk:]
            # ---- MLP backward
            da = _fgemm(dy, mlp, "wd_t", (mlp.down_proj.weight,), lambda: ops.transpose(wd))                                   # [T, Ip]
            if _fp8_fused(gu.shape[1]):     # the SwiGLU backward kernel emits the e4m3 operand of the dX contraction directly
                dq, dsc = ops.swiglu_bwd_quant(gu, da)
                del gu, da
                dh2 = _fgemm_q(dq, dsc, mlp, "wgu_t", (mlp.gate_proj.weight, mlp.up_proj.weight), lambda: ops.transpose(wgu))   # [T, H]
                del dq
            else:
                dgu = ops.swiglu_bwd(gu, da)
                del gu, da
                dh2 = _fgemm(dgu, mlp, "wgu_t", (mlp.gate_proj.weight, mlp.up_proj.weight), lambda: ops.transpose(wgu))            # [T, H]
                del dgu
            dx1 = ops.rmsnorm_bwd(x1, w2.weight, dh2, w2.variance_epsilon, add=dy)
            # ---- attention backward
            datt = _fgemm(dx1, at, "wo_t", (at.o_proj.weight,), lambda: ops.transpose(at.o_proj.weight.detach())).view(T, Hq, D)
            dqkv = torch.empty_like(qkv)
            ops.attn_varlen_bwd(q, k, v, att, datt, lse, cu, cu, max_len, max_len, D ** -0.5, True, dq=dqkv[:, :Hq], dk=dqkv[:, Hq:Hq + Hk],
                                dv=dqkv[:, Hq + Hk:])
            ops.rope_(dqkv, cos, _neg_table(sin), 0, Hq + Hk)   # inverse rotation (cos/sin tables are symmetric in the two halves)
            dqkv2 = dqkv.view(T, (Hq + 2 * Hk) * D)
            fold = (not _fp8["on"]) and T > 16 and _lora_cat_ok(at)
            dtt = None
            if fold:    # [dh1 | dt_q | dt_v] = dqkv [W | sB_q | sB_v] from one launch
                wqkv_t = _wt(at, "wqkv_t", at.q_proj.weight, at.k_proj.weight, at.v_proj.weight, build=lambda: ops.transpose(wqkv))
                dh1, dtt = ops.gemm_cat(dqkv2, wqkv_t, wn=_lora_cat_ops(at)[1])
            else:
                dh1 = _fgemm(dqkv2, at, "wqkv_t", (at.q_proj.weight, at.k_proj.weight, at.v_proj.weight), lambda: ops.transpose(wqkv))   # [T, H]
            grads = [None, None, None, None]
            if lq is not None or lv is not None:
                tn_pairs, tn_slots = [], []      # the layer's weight-gradient products (dB = dsl^T t_s, dA = dt^T dropout(h1) per LoRA module): ONE grouped launch below
                pend = []                        # (dt A, p, seed) of branches whose input gradient still has to pass its dropout mask on the way into dh1
                for slot, lp, lin, t_, cols, pdrop, hin, sd in ((0, lq, at.q_proj, tq, (0, Hq * D), pq, hq_in, 0),
                                                                (2, lv, at.v_proj, tv, ((Hq + Hk) * D, (Hq + 2 * Hk) * D), pv, hv_in, 1)):
                    if lp is None:
                        continue
                    _, At, Bts = _lora_ops(lin)                                         # A^T [H, r], (sB)^T [r, out]: once per step for all layers
                    dsl = dqkv2[:, cols[0]:cols[1]]                                    # [T, out]
                    if dtt is not None:
                        rr = dtt.shape[1] // 2
                        dt = dtt[:, :rr] if slot == 0 else dtt[:, rr:]                  # [T, r] = dsl (sB), from the folded product
                    else:
                        dt = ops.gemm(dsl, Bts)                                         # [T, r] = dsl (sB)
                    tn_pairs += [(dt, h1 if pdrop == 0.0 else hin), (dsl, t_)]          # dA [r, H] (lora_A saw the dropped input), dB [out, r] (t_s carries s)
                    tn_slots += [slot, slot + 1]
                    if pdrop == 0.0:
                        ops.gemm(dt, At, residual=dh1, out=dh1)                         # dh1 += dt A
                    else:                                                               # dh1 += mask / keep * (dt A): same seed, same mask
                        pend.append((ops.gemm(dt, At), pdrop, ctx.seeds[sd]))
                if len(pend) == 2 and dh1.is_contiguous():      # both branches: one launch, one rounding of dh1
                    ops.dropout_pair(pend[0][0], pend[0][1], pend[0][2], pend[1][0], pend[1][1], pend[1][2], accumulate_into=dh1)
                else:
                    for xg, pg, sg in pend:
                        ops.dropout(xg, pg, sg, out=dh1, accumulate=True)
                for sl, gw in zip(tn_slots, ops.gemm_tn_many(tn_pairs)):
                    grads[sl] = gw
            dx = ops.rmsnorm_bwd(x, w1.weight, dh1, w1.variance_epsilon, add=dx1)
        return (dx, grads[0], grads[1], grads[2], grads[3], None, None, None, None, None, None)


class EmbedFn(torch.autograd.Function):
    """x = embed_tokens[ids]; backward reduces the rows of duplicate ids by CSR segment sums (no atomics).  With a GradBucketReducer that registered
    the table as a sparse parameter (rga3.parallel.ddp) the gradient leaves as (unique row ids, summed rows) -- the 1.09 GB dense table gradient is never
    built, and data-parallel ranks exchange <= S rows each; otherwise the dense table gradient is returned to autograd."""

    @staticmethod
    def forward(ctx, weight, ids_dev, ids_np, grad_rows_np, plan=None):
        from ..parallel.ddp import sparse_sink_for

        ctx.shape = weight.shape
        ctx.sink = sparse_sink_for(weight)
        ctx.weight = weight
        plan = plan if plan is not None else {}
        if "embed_bwd" not in plan:     # CSR of duplicate ids, built (and uploaded) once per distinct input: host -> device copies are stream syncs
            rows = grad_rows_np                          # packed positions whose embedding came from the table (not vision placeholders)
            ids = ids_np[rows]
            order = np.argsort(ids, kind="stable")
            uniq, counts = np.unique(ids[order], return_counts=True)
            off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            dev = weight.device
            plan["embed_bwd"] = (upload(rows[order].astype(np.int64), dev), upload(off, dev), upload(uniq.astype(np.int64), dev),
                                 uniq.astype(np.int64))
        ctx.csr = plan["embed_bwd"]
        if ctx.sink is not None:
            ctx.sink.announce_sparse(weight, ctx.csr[3])
        return ops.gather_rows(weight, ids_dev)

    @staticmethod
    def backward(ctx, dx):
        rows_dev, off_dev, uniq_dev, _ = ctx.csr
        seg = ops.segment_sum_rows(dx.contiguous(), rows_dev, off_dev)
        if ctx.sink is not None:
            ctx.sink.add_sparse(ctx.weight, uniq_dev, seg)
            return None, None, None, None, None
        dW = torch.zeros(ctx.shape, dtype=dx.dtype, device=dx.device)
        ops.scatter_rows_(dW, uniq_dev, seg)
        return dW, None, None, None, None


class NormHeadCEFn(torch.autograd.Function):
    """loss = mean CE(lm_head(norm(h))[valid rows], labels): evaluates only rows with a label (value-identical to the reference's
    full-logits CE, HF ForCausalLMLoss).  Also returns the post-norm hidden states (no grad path through them yet)."""

    @staticmethod
    def forward(ctx, h, lm_w, norm_w, eps, valid_rows_dev, targets_dev, n_valid):
        hn = ops.rmsnorm(h, norm_w, eps)
        if n_valid == 0:
            ctx.n = 0
            ctx.save_for_backward(h, norm_w)
            ctx.eps = eps
            return torch.zeros((), device=h.device, dtype=torch.float32), hn
        hv = ops.gather_rows(hn, valid_rows_dev)
        logits = ops.gemm(hv, lm_w)
        row_loss, dlogits = ops.cross_entropy_rows(logits, targets_dev, want_grad=True, grad_scale=1.0 / n_valid)
        ctx.save_for_backward(h, lm_w, norm_w, hv, dlogits, valid_rows_dev)
        ctx.eps, ctx.n = eps, n_valid
        return row_loss.sum() / n_valid, hn

    @staticmethod
    def backward(ctx, gloss, ghn):
        if ctx.n == 0:   # no labelled row: only the hidden-state path (mask losses) carries gradient
            h, norm_w = ctx.saved_tensors
            with torch.no_grad():
                dh = ops.rmsnorm_bwd(h, norm_w, ghn.contiguous() if ghn is not None else torch.zeros_like(h), ctx.eps)
            return dh, None, None, None, None, None, None
        h, lm_w, norm_w, hv, dlogits, rows = ctx.saved_tensors
        n = ctx.n
        with torch.no_grad():
            dlogits = (dlogits.float() * gloss).to(dlogits.dtype)   # upstream gradient applied on the device ([labelled rows, V]: a few rows); no host read
            dW = None
            if lm_w.requires_grad:     # [V, H] = dlogits^T hv, written straight into the reducer's bucket slice when there is one (no 1.09 GB copy afterwards)
                from ..parallel.ddp import dense_grad_out_for
                dW = ops.gemm_tn(dlogits, hv, out=dense_grad_out_for(lm_w))
            # dhv [n, H] = dlogits [n, V] @ lm_w [V, H] as a TN product over the vocabulary (A = dlogits^T [V, n], B = lm_w as it lies): the NT form needs
            # lm_w^T, i.e. a 1.09 GB transpose of a TRAINABLE matrix every step (0.89 ms + the skinny product); here the table is streamed once
            n8 = (n + 7) // 8 * 8
            dl = dlogits
            if n8 != n:
                dl = torch.zeros((n8, dlogits.shape[1]), dtype=dlogits.dtype, device=dlogits.device)
                dl[:n] = dlogits
            dhv = ops.gemm_tn(ops.transpose(dl), lm_w.detach())[:n].contiguous()   # [n, H]
            dhn = ghn.contiguous().clone() if ghn is not None else torch.zeros_like(h)   # gradient arriving through hidden_states[-1]
            dhn_rows = ops.add(ops.gather_rows(dhn, rows), dhv)
            ops.scatter_rows_(dhn, rows, dhn_rows)
            dh = ops.rmsnorm_bwd(h, norm_w, dhn, ctx.eps)
        return dh, dW, None, None, None, None, None


def lm_train_forward(model, x, pos3, cu, max_len, labels_np, am_cur, flat_keep, lens, plan=None):
    """Decoder + LM head + CE with autograd nodes; x [T, H] packed embeddings (requires_grad if embed_tokens is trainable)."""
    tm = model.model
    cos, sin = tm.mrope_tables(pos3)
    lora_refresh(tm.layers)
    for li, layer in enumerate(tm.layers):
        at = layer.self_attn
        lq, lv = _lora_parts(at.q_proj), _lora_parts(at.v_proj)
        pq, pv = _lora_dropout(at)
        seeds = (next_dropout_seed(li, 0), next_dropout_seed(li, 1, advance=False)) if (pq > 0.0 or pv > 0.0) else None
        x = DecoderLayerFn.apply(x, lq[0] if lq else None, lq[1] if lq else None, lv[0] if lv else None, lv[1] if lv else None, layer, cos, sin,
                                 cu, max_len, seeds)
    plan = plan if plan is not None else {}
    if "train_targets" not in plan:     # labelled rows and their targets, uploaded once per distinct input
        B, S = labels_np.shape
        nxt = np.full((B, S), -100, dtype=np.int64)
        nxt[:, :-1] = labels_np[:, 1:]
        tgt = nxt.reshape(-1)[flat_keep]
        valid = np.flatnonzero(tgt != -100)
        dev = x.device
        plan["train_targets"] = (upload(valid.astype(np.int64), dev), upload(tgt[valid], dev), int(valid.size))
    valid_dev, tgt_dev, n_valid = plan["train_targets"]
    loss, hn = NormHeadCEFn.apply(x, model.lm_head.weight, tm.norm.weight, tm.norm.variance_epsilon, valid_dev, tgt_dev, n_valid)
    return loss, hn
