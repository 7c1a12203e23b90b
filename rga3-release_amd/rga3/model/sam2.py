"""SAM2 (Hiera + FPN image encoder, prompt encoder, two-way mask decoder, memory encoder / attention, video session)
on the rga3 HIP kernels — the build's own counterpart of reference model/sam2.py.

Public surface kept from the reference wrapper (model/sam2.py:87-446): ``SAM2(ckpt_path)`` with
``get_sam2_embeddings_train``, ``inject_language_embd_train``, ``get_sam2_embeddings``, ``language_embd_inference``
and a ``sam2_model`` attribute whose parameter names equal the reference's state dict, so
SAM2 checkpoints load and the trainer's substring selection (``sam_mask_decoder``, ``q_proj``, ``v_proj``) works.

MI355X-first design differences (outputs proven equal on fixtures, tests/test_sam2_gpu.py):
  * feature maps are token-major [pixels, C] end to end: every 1x1 conv / linear is the MFMA GEMM, LayerNorm2d is the
    row LayerNorm kernel, no NCHW<->NHWC permutes;
  * Hiera tokens live in window-major order for the stage's window size, so windowed attention is the varlen
    attention kernel on contiguous segments (no window_partition / unpartition copies); q-pooling is a max-pool kernel
    reading the Q slice of the fused QKV buffer through strides; layout changes (3 per image) are row gathers;
  * the best-IoU candidate is selected BEFORE the x4 bilinear upsample (reference upsamples all 3: sam2.py:3388-3402);
  * in the video session the image encoder runs once per frame (reference: twice, sam2.py:3539 one-entry cache).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip import autograd as AG
from ..hip import ops
from .qwen2_5_vl import Linear


_DEBUG = None   # set to a dict by tools/grad_locate.py (diagnostic)


_DECIMG = True    # mask decoder at inference: the image side of a two-way block boundary in one launch (csrc/decimg.hip)
_ROWCHAIN = True  # memory attention at inference: the row-wise steps between the attention kernels in one launch each (csrc/memlayer.hip)
_MLP_FUSE = True  # stage-1 / stage-2 MLP of the frozen Hiera trunk as one launch (csrc/hiera_mlp.hip); tools/ flip it for A/B runs
_MLP_FUSE_DIMS = (144, 288)
_LN_SUMS = True   # ... and its row statistics taken from the epilogue of the product that wrote the rows (rga3_gemm_lnsum_bf16 -> rga3_gemm_lnq_bf16) instead of a pass over them
_LN_FOLD = True   # LayerNorm of the frozen Hiera trunk folded into the consuming product (tools/ flip this module attribute for A/B runs; no environment switch)


def _ag():
    """Record autograd nodes (training of the mask decoder / text_hidden_fcs) instead of the fused inference kernels."""
    return torch.is_grad_enabled()


def _lin(l, x, residual=None, act="none"):
    if _ag():
        lora = getattr(l, "lora_A", None)
        if lora is None:
            return AG.linear(x, l.weight, l.bias, residual, act)
        # a LoRA-wrapped projection (the reference's target filter has a missing comma, train_joint.py:207-208, so PEFT also wraps the q_proj / v_proj
        # of SAM2's mask decoder and memory attention): base(x) + scaling * B(A(dropout(x))), gradients to A and B (and to the base weight if the
        # trainer's substring rule re-enabled it)
        if act != "none":
            raise NotImplementedError("LoRA on a projection with a fused activation")
        base = AG.linear(x, l.weight, l.bias, residual, "none")
        xin = x
        if l.training and l.dropout_p > 0.0:
            from .qwen_train import next_dropout_seed
            xin = AG.DropoutFn.apply(x.contiguous(), l.dropout_p, next_dropout_seed(0, 2))
        t = AG.linear(xin, l.lora_A["default"].weight)
        return AG.linear(t, l.lora_B["default"].weight * l.scaling, None, base)
    return l(x, residual=residual, act=act)


# ------------------------------------------------------------------------------------------------ parameter holders
class ConvParams(nn.Module):
    """Holds a conv's weight/bias under the reference's names; the arithmetic runs in the HIP kernels."""

    def __init__(self, cout, cin, kh, kw, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, kh, kw))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def as_linear(self):
        return self.weight.reshape(self.weight.shape[0], -1)


class NormParams(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x2d, act="none"):
        if _ag():
            y = AG.LayerNormFn.apply(x2d.contiguous(), self.weight, self.bias, self.eps)
            return AG.GeluFn.apply(y) if act == "gelu" else y
        return ops.layernorm(x2d, self.weight, self.bias, self.eps, act=act)


class MLP(nn.Module):
    """reference sam2.py:2305-2329"""

    def __init__(self, i, h, o, n, sigmoid_output=False, act="relu"):
        super().__init__()
        hs = [h] * (n - 1)
        self.layers = nn.ModuleList(Linear(a, b) for a, b in zip([i] + hs, hs + [o]))
        self.num_layers, self.sigmoid_output, self.act = n, sigmoid_output, act

    def forward(self, x, residual=None):
        for i, l in enumerate(self.layers):
            last = i == self.num_layers - 1
            if _ag() and not last and self.act == "gelu":
                x = AG.GeluFn.apply(_lin(l, x))
            else:
                x = _lin(l, x, act="none" if last else self.act, residual=residual if last else None)
        return torch.sigmoid(x.float()).to(x.dtype) if self.sigmoid_output else x


# ------------------------------------------------------------------------------------------------ layouts
_PERM_CACHE: Dict[tuple, torch.Tensor] = {}


def _layout_order(H, W, w):
    """raster index of each position of the window-major(w) layout (w == 0: raster)."""
    if w == 0:
        return np.arange(H * W)
    idx = np.arange(H * W).reshape(H // w, w, W // w, w).transpose(0, 2, 1, 3).reshape(-1)
    return idx


def relayout(x, F, H, W, w_from, w_to):
    """Reorder token rows of x [F*H*W, C] between window-major layouts (row gather with a cached permutation)."""
    if w_from == w_to:
        return x
    key = (F, H, W, w_from, w_to, str(x.device))
    if key not in _PERM_CACHE:
        a, b = _layout_order(H, W, w_from), _layout_order(H, W, w_to)
        inv_a = np.empty_like(a)
        inv_a[a] = np.arange(a.size)
        p = inv_a[b]
        full = (p[None, :] + (np.arange(F) * H * W)[:, None]).reshape(-1)
        _PERM_CACHE[key] = torch.from_numpy(full.astype(np.int64)).to(x.device)
    return ops.gather_rows(x, _PERM_CACHE[key])


_CU_CACHE: Dict[tuple, torch.Tensor] = {}


def _cu(nseg, seglen, device):
    key = (nseg, seglen, str(device))
    if key not in _CU_CACHE:
        _CU_CACHE[key] = (torch.arange(nseg + 1, dtype=torch.int64) * seglen).to(torch.int32).to(device)
    return _CU_CACHE[key]


# ------------------------------------------------------------------------------------------------ Hiera trunk
class PatchEmbed(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.proj = ConvParams(embed_dim, 3, 7, 7)


class MultiScaleAttention(nn.Module):
    def __init__(self, dim, dim_out, heads):
        super().__init__()
        self.qkv = Linear(dim, dim_out * 3)
        self.proj = Linear(dim_out, dim_out)
        self.num_heads = heads


class MultiScaleBlock(nn.Module):
    """reference sam2.py:1035-1117 on window-major tokens."""

    def __init__(self, dim, dim_out, heads, window, pool):
        super().__init__()
        self.dim, self.dim_out, self.window, self.pool_q = dim, dim_out, window, pool
        self.norm1 = NormParams(dim, 1e-6)
        self.attn = MultiScaleAttention(dim, dim_out, heads)
        self.norm2 = NormParams(dim_out, 1e-6)
        self.mlp = MLP(dim_out, dim_out * 4, dim_out, 2, act="gelu")
        if dim != dim_out:
            self.proj = Linear(dim, dim_out)
        self._fold_cache = {}

    def _folded(self, which):
        """(Wf, colc, biasf) of a projection with the preceding LayerNorm folded in (ops.fold_layernorm), rebuilt when either module's parameters change."""
        lin, norm = {"qkv": (self.attn.qkv, self.norm1), "proj": (getattr(self, "proj", None), self.norm1), "fc1": (self.mlp.layers[0], self.norm2)}[which]
        key = (lin.weight.data_ptr(), lin.weight._version, None if lin.bias is None else lin.bias._version, norm.weight.data_ptr(), norm.weight._version, norm.bias._version)
        c = self._fold_cache.get(which)
        if c is None or c[0] != key:
            c = self._fold_cache[which] = (key, ops.fold_layernorm(lin.weight, lin.bias, norm.weight, norm.bias))
        return c[1]

    def forward(self, x, Fn, H, W, layout_w, sums=None):
        """x [Fn*H*W, dim] in window-major(layout_w) order. Returns (x, H, W, layout_w).
        sums (frozen trunk only): a _SumsFeed -- `sums.of_x` holds the per-tile-column (sum, sum of squares) of the rows of x if the product that wrote x left them
        (ops.gemm_lnsum); on return it holds those of the block's output (or None).  The LayerNorm statistics then never cost a pass over the rows."""
        ws = self.window
        if ws > 0 and (H % ws or W % ws):
            raise NotImplementedError("Hiera window padding (map not divisible by the window) is not on the RGA3 path (images are 1024x1024)")
        if ws > 0:
            if layout_w != ws and sums is not None:
                sums.of_x = None          # the rows move: the producer's per-row sums no longer line up
            x = relayout(x, Fn, H, W, layout_w, ws)
            layout_w = ws
        heads, do = self.attn.num_heads, self.dim_out
        # frozen trunk (no autograd): both LayerNorms are folded into the products that consume them -- a row-statistics pass (one read of x) and a
        # gamma-folded weight replace the LayerNorm pass (read + write) and the re-read of its output (rga3_gemm_ln_bf16)
        fold = _LN_FOLD and not _ag() and x.is_contiguous() and self.dim % 8 == 0
        feed = sums if (fold and sums is not None and _LN_SUMS) else None
        if feed is not None and feed.of_x is not None:
            st1 = ops.LnSums(feed.of_x, self.norm1.eps)       # left by the previous block's fc2 epilogue
        else:
            st1 = ops.layernorm_stats(x, self.norm1.eps) if fold else None
        if sums is not None:
            sums.of_x = None
        h = None if fold else self.norm1(x)
        T = H * W
        nwin = Fn * (T // (ws * ws)) if ws > 0 else Fn
        seg = ws * ws if ws > 0 else T
        shortcut = x
        if self.dim != self.dim_out:
            shortcut = ops.gemm_ln(x, st1, *self._folded("proj")) if fold else self.proj(h)
        if self.pool_q:
            if ws == 0:
                raise NotImplementedError("q-pooling inside a global-attention block does not occur in Hiera configs used by SAM2")
            if self.dim != self.dim_out:
                shortcut = ops.maxpool2x2_win(shortcut, nwin, ws)
        qkv = ops.gemm_ln(x, st1, *self._folded("qkv")) if fold else self.attn.qkv(h)   # [N, 3*do]
        q = qkv[:, :do]
        seg_q = seg
        if self.pool_q:
            q = ops.maxpool2x2_win(q, nwin, ws)                 # strided read of the Q slice
            seg_q = seg // 4
        hd = do // heads
        qv = q.view(q.shape[0], heads, hd) if q.is_contiguous() else q.unflatten(1, (heads, hd))
        kv = qkv.view(qkv.shape[0], 3 * heads, hd)
        # tiny windows (16 tokens in stage 2, 4 pooled queries x 16 keys at the stage change): g windows are packed into one segment of up to
        # 64 query rows with block-diagonal visibility -- one workgroup per 64 rows instead of one per window
        g = 1
        if ws > 0 and seg_q < 64 and (seg_q & (seg_q - 1)) == 0 and (seg & (seg - 1)) == 0:
            while seg_q * g * 2 <= 64 and nwin % (g * 2) == 0:
                g *= 2
        if g > 1:
            att = ops.attn_varlen(qv, kv[:, heads:2 * heads], kv[:, 2 * heads:], _cu(nwin // g, seg_q * g, x.device), _cu(nwin // g, seg * g, x.device),
                                  seg_q * g, hd ** -0.5, causal=False, block=(seg_q, seg), max_k=seg * g)
        else:   # max_k: windows of <= 256 keys take the whole-segment-in-LDS kernel
            att = ops.attn_varlen(qv, kv[:, heads:2 * heads], kv[:, 2 * heads:], _cu(nwin, seg_q, x.device), _cu(nwin, seg, x.device), seg_q,
                                  hd ** -0.5, causal=False, max_k=seg)
        l0, l1 = self.mlp.layers[0], self.mlp.layers[1]
        fused_mlp = (fold and _MLP_FUSE and self.dim_out in _MLP_FUSE_DIMS and l0.out_features == 4 * self.dim_out and self.mlp.num_layers == 2 and self.mlp.act == "gelu"
                     and l0.bias is not None and l1.bias is not None)
        # the two residual-writing products leave the statistics of the rows they write (rga3_gemm_lnsum_bf16): proj's feed norm2 -> fc1, fc2's the next block's norm1
        lnq = feed is not None and not fused_mlp and self.dim_out % 8 == 0 and att.shape[0] > 16
        pr = self.attn.proj
        if lnq:
            x, s2 = ops.gemm_lnsum(att.view(att.shape[0], do), pr.weight, pr.bias, residual=shortcut)
        else:
            x = pr(att.view(att.shape[0], do), residual=shortcut)
        if self.pool_q:
            H, W, layout_w = H // 2, W // 2, ws // 2
        if fused_mlp and x.is_contiguous():
            # stage-1 / stage-2 MLP (144 -> 576 -> 144 over 65 536 tokens per frame, 288 -> 1152 -> 288 over 16 384) in one launch: the hidden activation and the
            # LayerNorm statistics never reach HBM
            x = ops.hiera_mlp(x, *self._folded("fc1"), l1.weight, l1.bias, self.norm2.eps)
        elif lnq:
            hmid = ops.gemm_ln(x, ops.LnSums(s2, self.norm2.eps), *self._folded("fc1"), act="gelu")
            x, feed.of_x = ops.gemm_lnsum(hmid, l1.weight, l1.bias, residual=x)
        elif fold and self.dim_out % 8 == 0:
            hmid = ops.gemm_ln(x, ops.layernorm_stats(x, self.norm2.eps), *self._folded("fc1"), act="gelu")
            x = self.mlp.layers[1](hmid, residual=x)
        else:
            x = self.mlp(self.norm2(x), residual=x)
        return x, H, W, layout_w


class _SumsFeed:
    """`of_x`: the LayerNorm partial sums (ops.gemm_lnsum) of the rows currently flowing between two blocks of one frozen trunk pass, or None."""

    def __init__(self):
        self.of_x = None


class Hiera(nn.Module):
    def __init__(self, embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43), window_spec=(8, 4, 16, 8),
                 window_pos_embed_bkg_spatial_size=(7, 7), q_pool=3):
        super().__init__()
        depth = sum(stages)
        self.stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
        self.q_pool_blocks = [x + 1 for x in self.stage_ends[:-1]][:q_pool]
        self.patch_embed = PatchEmbed(embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, *window_pos_embed_bkg_spatial_size))
        self.pos_embed_window = nn.Parameter(torch.zeros(1, embed_dim, window_spec[0], window_spec[0]))
        self.blocks = nn.ModuleList()
        cur_stage, dim, heads = 1, embed_dim, num_heads
        for i in range(depth):  # construction rule of reference sam2.py:1182-1210 (window lags the stage change by one block)
            dim_out = dim
            window = window_spec[cur_stage - 1]
            if i in global_att_blocks:
                window = 0
            if i - 1 in self.stage_ends:
                dim_out, heads = dim * 2, heads * 2
                cur_stage += 1
            self.blocks.append(MultiScaleBlock(dim, dim_out, heads, window, i in self.q_pool_blocks))
            dim = dim_out
        self.channel_list = [self.blocks[i].dim_out for i in self.stage_ends[::-1]]
        self._pos_cache = {}

    def pos_tokens(self, h, w):
        """bicubic-interpolated background pos-embed + tiled window embed as raster tokens [h*w, C] (sam2.py:1218-1226)."""
        key = (h, w, self.pos_embed._version, self.pos_embed.device)
        if key not in self._pos_cache:
            with torch.no_grad():
                pe = F.interpolate(self.pos_embed.float(), size=(h, w), mode="bicubic")
                we = self.pos_embed_window.float()
                pe = pe + we.tile([a // b for a, b in zip(pe.shape, we.shape)])
                self._pos_cache = {key: pe[0].permute(1, 2, 0).reshape(h * w, -1).to(self.pos_embed.dtype).contiguous()}
        return self._pos_cache[key]

    def forward(self, img):
        """img [Fn, 3, S, S] bf16 -> list of (tokens [Fn*H*W, C] raster order, H, W) per stage (high-res first)."""
        Fn = img.shape[0]
        cols, (H, W) = ops.im2col(img.contiguous(), 7, 4, 3)
        wp = self.patch_embed.proj.as_linear()
        if wp.shape[1] != cols.shape[1]:
            wp = ops.pad_cols(wp.contiguous(), cols.shape[1])
        x = ops.gemm(cols, wp, self.patch_embed.proj.bias)
        x = ops.add_bcast(x, self.pos_tokens(H, W))
        layout = 0
        outs = []
        # frozen pass: the blocks hand the LayerNorm statistics of their outputs forward
        feed = _SumsFeed() if (not _ag() and _LN_SUMS and x.is_cuda) else None
        for i, blk in enumerate(self.blocks):
            x, H, W, layout = blk(x, Fn, H, W, layout, feed)
            if i in self.stage_ends:
                outs.append((x, H, W, layout))
        return outs


# ------------------------------------------------------------------------------------------------ position encodings
def position_embedding_sine(num_pos_feats, h, w, device, temperature=10000.0):
    """reference sam2.py:1781-1814 (normalize=True, scale=2*pi) -> raster tokens [h*w, num_pos_feats] f32 (host math, cached by callers)."""
    npf = num_pos_feats // 2
    y = torch.arange(1, h + 1, dtype=torch.float32).view(-1, 1).repeat(1, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, -1).repeat(h, 1)
    y = y / (y[-1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = temperature ** (2 * (torch.arange(npf, dtype=torch.float32) // 2) / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).reshape(h * w, -1).to(device)


class FpnNeck(nn.Module):
    def __init__(self, d_model, backbone_channel_list, fpn_top_down_levels=(2, 3)):
        super().__init__()
        self.d_model = d_model
        self.backbone_channel_list = list(backbone_channel_list)
        self.convs = nn.ModuleList()
        for c in backbone_channel_list:
            seq = nn.Sequential()
            seq.add_module("conv", ConvParams(d_model, c, 1, 1))
            self.convs.append(seq)
        self.fpn_top_down_levels = list(fpn_top_down_levels)
        self._pos = {}

    def pos(self, h, w, device, dtype):
        key = (h, w, str(device), dtype)
        if key not in self._pos:
            self._pos[key] = position_embedding_sine(self.d_model, h, w, device).to(dtype)
        return self._pos[key]

    def forward(self, stages, Fn):
        """stages: trunk outputs (high-res first). Returns raster-token maps per level [(tokens, H, W)], high-res first."""
        n = len(self.convs) - 1
        out = [None] * len(stages)
        prev = None
        for i in range(n, -1, -1):
            x, H, W, layout = stages[i]
            conv = self.convs[n - i].conv
            lat = ops.gemm(x, conv.as_linear(), conv.bias)
            lat = relayout(lat, Fn, H, W, layout, 0)
            if i in self.fpn_top_down_levels and prev is not None:
                prev = ops.upsample2x_add(lat, prev, Fn, H, W)
            else:
                prev = lat
            out[i] = (prev, H, W)
        return out


class ImageEncoder(nn.Module):
    def __init__(self, trunk, neck, scalp=1):
        super().__init__()
        self.trunk, self.neck, self.scalp = trunk, neck, scalp


# ------------------------------------------------------------------------------------------------ SAM heads
class Attention(nn.Module):
    """reference sam2.py:1417-1481 (projections as GEMMs, softmax(QK^T)V as the varlen kernel, batch = segments)."""

    def __init__(self, embedding_dim, num_heads, downsample_rate=1, kv_in_dim=None):
        super().__init__()
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        kv = kv_in_dim if kv_in_dim is not None else embedding_dim
        self.q_proj = Linear(embedding_dim, self.internal_dim)
        self.k_proj = Linear(kv, self.internal_dim)
        self.v_proj = Linear(kv, self.internal_dim)
        self.out_proj = Linear(self.internal_dim, embedding_dim)

    def forward(self, q, k, v, B, nq, nk, residual=None):
        """q [B*nq, C], k/v [B*nk, Ckv] -> out_proj(attn) (+ residual) [B*nq, C]."""
        H, hd = self.num_heads, self.internal_dim // self.num_heads
        qp, kp, vp = _lin(self.q_proj, q), _lin(self.k_proj, k), _lin(self.v_proj, v)
        if _ag():
            o = AG.AttnFn.apply(qp.view(-1, H, hd), kp.view(-1, H, hd), vp.view(-1, H, hd), _cu(B, nq, q.device), _cu(B, nk, q.device), nq, nk, hd ** -0.5)
        else:
            o = ops.attn_varlen(qp.view(-1, H, hd), kp.view(-1, H, hd), vp.view(-1, H, hd), _cu(B, nq, q.device), _cu(B, nk, q.device), nq, hd ** -0.5, max_k=nk)
        return _lin(self.out_proj, o.reshape(-1, self.internal_dim), residual=residual)

    def attend(self, qp, kp, vp, B, nq, nk, residual=None, vt=False, vbias=None, k_head_major=False):
        """softmax(QK^T)V + out_proj on projections that already exist (qp [B*nq, internal_dim], kp / vp [B*nk, internal_dim]; vt: vp is the TRANSPOSED values
        [B * internal_dim, nk] for the few-query kernel, vbias its bias if not yet added)."""
        H, hd = self.num_heads, self.internal_dim // self.num_heads
        if vt:
            o = ops.attn_fewq(qp, kp, vp, nq, nk, H, hd ** -0.5, vbias, k_head_major=k_head_major)
            return _lin(self.out_proj, o, residual=residual)
        o = ops.attn_varlen(qp.view(-1, H, hd), kp.view(-1, H, hd), vp.view(-1, H, hd), _cu(B, nq, qp.device), _cu(B, nk, qp.device), nq, hd ** -0.5, max_k=nk)
        return _lin(self.out_proj, o.reshape(-1, self.internal_dim), residual=residual)


def _add(a, b):
    return AG.AddFn.apply(a, b) if _ag() else ops.add(a, b)


def _add_bcast(a, b, alpha=1.0):
    return AG.AddBcastFn.apply(a, b, alpha) if _ag() else ops.add_bcast(a, b, alpha)


class TwoWayAttentionBlock(nn.Module):
    def __init__(self, dim, heads, mlp_dim, skip_first_layer_pe):
        super().__init__()
        self.self_attn = Attention(dim, heads)
        self.norm1 = NormParams(dim, 1e-5)
        self.cross_attn_token_to_image = Attention(dim, heads, downsample_rate=2)
        self.norm2 = NormParams(dim, 1e-5)
        self.mlp = MLP(dim, mlp_dim, dim, 2)
        self.norm3 = NormParams(dim, 1e-5)
        self.norm4 = NormParams(dim, 1e-5)
        self.cross_attn_image_to_token = Attention(dim, heads, downsample_rate=2)
        self.skip_first_layer_pe = skip_first_layer_pe

    def forward(self, queries, keys, query_pe, key_pe, B, nq, nk):
        if self.skip_first_layer_pe:
            queries = self.self_attn(queries, queries, queries, B, nq, nq)
        else:
            q = _add(queries, query_pe)
            queries = self.self_attn(q, q, queries, B, nq, nq, residual=queries)
        queries = self.norm1(queries)
        q = _add(queries, query_pe)
        k = _add_bcast(keys, key_pe)
        queries = self.norm2(self.cross_attn_token_to_image(q, k, keys, B, nq, nk, residual=queries))
        queries = self.norm3(self.mlp(queries, residual=queries))
        q = _add(queries, query_pe)
        keys = self.norm4(self.cross_attn_image_to_token(k, q, queries, B, nk, nq, residual=keys))
        return queries, keys


    def forward_fused(self, queries, keys, query_pe, key_pe, B, nq, nk, kp, vp, nxt):
        """Inference form: the token side as above; the image side of the block's end -- image-to-token attention, norm4 and the NEXT token-to-image attention's
        k / v projections -- in one launch (csrc/decimg.hip).  kp / vp: this block's token-to-image keys / values if the previous block already produced them."""
        # token side: the projections of one attention are independent products of (tokens + positional tokens) / tokens -- one launch per attention, the sum
        # taken while the rows are loaded (csrc/gemm_bf16.hip gemm_rows16_many_kernel)
        sa, t2i, i2t = self.self_attn, self.cross_attn_token_to_image, self.cross_attn_image_to_token
        lin = lambda m_: (m_.weight, m_.bias)
        pe_ = None if self.skip_first_layer_pe else query_pe
        qp, kp_, vp_ = ops.gemm_rows16_many([(queries, pe_) + lin(sa.q_proj), (queries, pe_) + lin(sa.k_proj), (queries, None) + lin(sa.v_proj)])
        queries = sa.attend(qp, kp_, vp_, B, nq, nq, residual=None if self.skip_first_layer_pe else queries)
        queries = self.norm1(queries)
        fewq = B == 1 and nk <= 4096 and nk % 16 == 0      # the few-query attention kernel (values transposed, no merge launch: csrc/decimg.hip)
        vbias = None
        if kp is None:
            kp = t2i.k_proj(_add_bcast(keys, key_pe))
            if fewq:
                vp, vbias = ops.gemm(t2i.v_proj.weight, keys), t2i.v_proj.bias      # W_v keys^T = v^T [128, nk]; the bias goes in after the softmax
            else:
                vp = t2i.v_proj(keys)
        (qp,) = ops.gemm_rows16_many([(queries, query_pe) + lin(t2i.q_proj)])
        queries = self.norm2(t2i.attend(qp, kp, vp, B, nq, nk, residual=queries, vt=fewq, vbias=vbias, k_head_major=fewq and vbias is None))
        queries = self.norm3(self.mlp(queries, residual=queries))
        kt, vt = ops.gemm_rows16_many([(queries, query_pe) + lin(i2t.k_proj), (queries, None) + lin(i2t.v_proj)])
        hd = i2t.internal_dim // i2t.num_heads
        keys, kp2, vp2 = ops.decimg_rows(keys, key_pe, kt, vt, nq, (i2t.q_proj.weight, i2t.q_proj.bias),
                                         (i2t.out_proj.weight, i2t.out_proj.bias), (self.norm4.weight, self.norm4.bias), self.norm4.eps,
                                         (nxt.k_proj.weight, nxt.k_proj.bias), (nxt.v_proj.weight, nxt.v_proj.bias), scale=hd ** -0.5, v_transposed=fewq)
        return queries, keys, kp2, vp2


class TwoWayTransformer(nn.Module):
    def __init__(self, depth, dim, heads, mlp_dim):
        super().__init__()
        self.layers = nn.ModuleList(TwoWayAttentionBlock(dim, heads, mlp_dim, i == 0) for i in range(depth))
        self.final_attn_token_to_image = Attention(dim, heads, downsample_rate=2)
        self.norm_final_attn = NormParams(dim, 1e-5)

    def _fusable(self, keys, key_pe, nq, B=1):
        """The one-launch image side (csrc/decimg.hip): inference, plain projections, SAM2's decoder geometry (256 wide, 128 internal = 8 heads x 16)."""
        a = self.final_attn_token_to_image
        mods = [a.k_proj, a.v_proj]
        for l in self.layers:
            mods += [l.cross_attn_image_to_token.q_proj, l.cross_attn_image_to_token.out_proj, l.cross_attn_token_to_image.k_proj, l.cross_attn_token_to_image.v_proj]
        for l in self.layers:
            mods += [l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj, l.cross_attn_token_to_image.q_proj, l.cross_attn_image_to_token.k_proj,
                     l.cross_attn_image_to_token.v_proj]
        mods.append(a.q_proj)
        return (_DECIMG and not _ag() and keys.shape[1] == 256 and a.internal_dim == 128 and a.num_heads == 8 and B * nq <= 16 and key_pe.shape[0] >= 16
                and all(l.cross_attn_image_to_token.internal_dim == 128 and l.cross_attn_image_to_token.num_heads == 8 for l in self.layers)
                and not any(hasattr(m_, "lora_A") for m_ in mods))

    def forward(self, keys, key_pe, tokens, B, nq, nk):
        queries, query_pe = tokens, tokens
        if self._fusable(keys, key_pe, nq, B):
            kp = vp = None          # the image-side k / v of the next token-to-image attention, produced by the previous block's fused tail
            for li, layer in enumerate(self.layers):
                nxt = self.layers[li + 1].cross_attn_token_to_image if li + 1 < len(self.layers) else self.final_attn_token_to_image
                queries, keys, kp, vp = layer.forward_fused(queries, keys, query_pe, key_pe, B, nq, nk, kp, vp, nxt)
            fa = self.final_attn_token_to_image
            (qp,) = ops.gemm_rows16_many([(queries, query_pe, fa.q_proj.weight, fa.q_proj.bias)])
            fq = B == 1 and nk <= 4096 and nk % 16 == 0
            queries = self.norm_final_attn(fa.attend(qp, kp, vp, B, nq, nk, residual=queries, vt=fq, k_head_major=fq))
            return queries, keys
        for layer in self.layers:
            queries, keys = layer(queries, keys, query_pe, key_pe, B, nq, nk)
        q = _add(queries, query_pe)
        k = _add_bcast(keys, key_pe)
        queries = self.norm_final_attn(self.final_attn_token_to_image(q, k, keys, B, nq, nk, residual=queries))
        return queries, keys


class ConvTParams(nn.Module):
    """ConvTranspose2d(k=2, s=2) parameters [Cin, Cout, 2, 2]; as_linear() gives the GEMM weight [(dy,dx,co), ci]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, 2, 2))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self._pk = None

    def as_linear(self):
        key = (self.weight.data_ptr(), self.weight._version, self.weight.dtype)
        if self._pk is None or self._pk[0] != key:
            self._pk = (key, self.weight.detach().permute(2, 3, 1, 0).reshape(-1, self.weight.shape[0]).contiguous())
        return self._pk[1]

    def as_linear_grad(self):
        """same matrix, recorded by autograd (the permute/reshape are views + one copy that torch differentiates)."""
        return self.weight.permute(2, 3, 1, 0).reshape(-1, self.weight.shape[0]).contiguous()


class _HeadsOut(dict):
    """forward_sam_heads' result on the inference path: the reference's keys, with "ious" (the multimask IoU predictions, reference sam2.py:3360-3373 / :3398) made on first read."""

    def __init__(self, d, iou):
        super().__init__(d)
        self._iou = iou

    def __missing__(self, key):
        if key != "ious":
            raise KeyError(key)
        v = self._iou[:, 1:].float()
        self[key] = v
        return v

    def get(self, key, default=None):
        return self[key] if key == "ious" else super().get(key, default)

    def __contains__(self, key):
        return key == "ious" or super().__contains__(key)


class MaskDecoder(nn.Module):
    """reference sam2.py:1926-2160 with pred_obj_scores(+mlp), high-res features, multimask tokens for the object pointer."""

    def __init__(self, dim, depth=2, heads=8, mlp_dim=2048, iou_head_hidden_dim=256):
        super().__init__()
        self.transformer_dim = dim
        self.transformer = TwoWayTransformer(depth, dim, heads, mlp_dim)
        self.iou_token = nn.Embedding(1, dim)
        self.num_mask_tokens = 4
        self.mask_tokens = nn.Embedding(4, dim)
        self._tok_cache = None       # (key, tokens) of forward(..., const_sparse=True)
        self.obj_score_token = nn.Embedding(1, dim)
        self.output_upscaling = nn.Sequential(ConvTParams(dim, dim // 4), NormParams(dim // 4, 1e-6), nn.GELU(), ConvTParams(dim // 4, dim // 8), nn.GELU())
        self.conv_s0 = ConvParams(dim // 8, dim, 1, 1)
        self.conv_s1 = ConvParams(dim // 4, dim, 1, 1)
        self.output_hypernetworks_mlps = nn.ModuleList(MLP(dim, dim, dim // 8, 3) for _ in range(4))
        self.iou_prediction_head = MLP(dim, iou_head_hidden_dim, 4, 3, sigmoid_output=True)
        self.pred_obj_score_head = MLP(dim, dim, 1, 3)

    def forward(self, pix_tokens, pos_tokens, sparse, feat_s0, feat_s1, B, h, w, const_sparse=False):
        """pix_tokens [B*h*w, C] (already + dense prompt), pos_tokens [h*w, C], sparse [B, n, C] (const_sparse: it is the prompt encoder's padding alone).
        Returns masks f32 [B, 4, 4h, 4w], iou [B, 4], mask tokens [B, 4, C], object score logits [B, 1]."""
        C = self.transformer_dim
        tokens = None
        if const_sparse and not _ag():
            # tracked frames carry no prompt: output tokens + the two padding points are a function of the weights alone -- built once, not with two concatenation
            # launches per frame (a frame of the stream is ~115 dependent launches of >= 4.4 us each)
            srcs = (self.obj_score_token.weight, self.iou_token.weight, self.mask_tokens.weight, sparse)
            key = tuple((t_.data_ptr(), t_._version, tuple(t_.shape), t_.dtype) for t_ in srcs) + (B,)
            if self._tok_cache is not None and self._tok_cache[0] == key:
                tokens = self._tok_cache[1]
        if tokens is None:
            out_tokens = torch.cat([self.obj_score_token.weight, self.iou_token.weight, self.mask_tokens.weight], dim=0)
            tokens = torch.cat([out_tokens[None].expand(B, -1, -1), sparse.to(out_tokens.dtype)], dim=1).contiguous()
            if const_sparse and not _ag():
                self._tok_cache = (key, tokens.detach())
        nq = tokens.shape[1]
        hs, src = self.transformer(pix_tokens, pos_tokens, tokens.view(B * nq, C), B, nq, h * w)
        hs = hs.view(B, nq, C)
        iou_tok, mask_toks = hs[:, 1], hs[:, 2:6]
        dc1, ln1, _, dc2, _ = self.output_upscaling
        npx = 16 * h * w
        if _ag():
            g1 = AG.linear(src, dc1.as_linear_grad())
            up = AG.PixelShuffleFn.apply(g1, dc1.bias, feat_s1, B, h, w)
            up = ln1(up, act="gelu")
            g2 = AG.linear(up, dc2.as_linear_grad())
            up = AG.GeluFn.apply(AG.PixelShuffleFn.apply(g2, dc2.bias, feat_s0, B, 2 * h, 2 * w))
            hyper = torch.stack([self.output_hypernetworks_mlps[i](mask_toks[:, i].contiguous()) for i in range(4)], dim=1)
            masks = AG.MaskProductFn.apply(hyper, up, npx).view(B, 4, 4 * h, 4 * w)
            if _DEBUG is not None:   # tools/grad_locate.py: keep the gradients of the head's intermediates
                _DEBUG.update(hs=hs, src=src, up=up, hyper=hyper, masks=masks, tokens=tokens)
                for t_ in (hs, src, up, hyper, masks, tokens):
                    if t_.requires_grad:
                        t_.retain_grad()
        else:
            g1 = ops.gemm(src, dc1.as_linear())
            up = ops.pixel_shuffle2x(g1, dc1.bias, feat_s1, B, h, w)
            up = ln1(up, act="gelu")
            g2 = ops.gemm(up, dc2.as_linear())
            up = ops.pixel_shuffle2x(g2, dc2.bias, feat_s0, B, 2 * h, 2 * w, act="gelu")          # [B*16hw, C/8]
            heads = self._fused_heads(hs, B, nq)
            if heads is not None:     # the six 3-layer MLPs on single token rows in ONE launch (csrc/dechead.hip)
                hyper, iou, obj = heads
                masks = ops.mask_product(hyper, up, npx).view(B, 4, 4 * h, 4 * w)
                return masks, iou, mask_toks, obj
            hyper = torch.stack([self.output_hypernetworks_mlps[i](mask_toks[:, i].contiguous()) for i in range(4)], dim=1)  # [B, 4, C/8]
            masks = ops.mask_product(hyper.contiguous(), up, npx).view(B, 4, 4 * h, 4 * w)   # masks[b] = hyper[b] @ up[b]^T, planes row-major, f32
        iou = self.iou_prediction_head(iou_tok.contiguous())
        obj = self.pred_obj_score_head(hs[:, 0].contiguous())
        return masks, iou, mask_toks, obj

    def _fused_heads(self, hs, B, nq):
        """hyper [B, 4, C/8], iou [B, 4] (sigmoid), obj [B, 1] from the decoder's output tokens hs [B, nq, C]: one launch.  None when a head is LoRA-wrapped or has an
        unusual shape (the per-layer path above then runs)."""
        C = self.transformer_dim
        mlps = list(self.output_hypernetworks_mlps) + [self.iou_prediction_head, self.pred_obj_score_head]
        if C % 8 or C > 512 or any(m_.num_layers != 3 or m_.act != "relu" or any(hasattr(l, "lora_A") for l in m_.layers) or m_.layers[0].out_features % 8
                                   or m_.layers[0].out_features > 512 for m_ in mlps):
            return None
        if self.iou_prediction_head.layers[2].out_features != 4 or not self.iou_prediction_head.sigmoid_output:
            return None
        hs = hs.contiguous()
        flat = hs.view(-1)
        specs = []
        hyper = torch.empty((B, 4, mlps[0].layers[2].out_features), dtype=hs.dtype, device=hs.device)
        for i, m_ in enumerate(mlps):
            tok = (2 + i) if i < 4 else (1 if i == 4 else 0)      # mask tokens 2..5, IoU token 1, object-score token 0
            w = tuple(t for l in m_.layers for t in (l.weight, l.bias))
            specs.append((flat[tok * C:], nq * C, w, i == 4, hyper[:, i] if i < 4 else None))
        outs = ops.mlp3_rows(specs, B)
        return hyper, outs[4], outs[5]


class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats):
        super().__init__()
        self.register_buffer("positional_encoding_gaussian_matrix", torch.randn(2, num_pos_feats))

    def dense_tokens(self, h, w):
        """reference sam2.py:1832-1856 -> raster tokens [h*w, 2F] f32"""
        G = self.positional_encoding_gaussian_matrix.float()
        grid = torch.ones((h, w), dtype=torch.float32, device=G.device)
        y, x = (grid.cumsum(0) - 0.5) / h, (grid.cumsum(1) - 0.5) / w
        # a 2-term contraction written out (x' G[0] + y' G[1]): a torch `@` here would be a vendor-BLAS launch on the product path
        c = 2 * math.pi * ((2 * x - 1)[..., None] * G[0] + (2 * y - 1)[..., None] * G[1])
        return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).reshape(h * w, -1)


class PromptEncoder(nn.Module):
    def __init__(self, embed_dim, image_embedding_size, input_image_size, mask_in_chans=16):
        super().__init__()
        self.embed_dim, self.image_embedding_size, self.input_image_size = embed_dim, image_embedding_size, input_image_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.point_embeddings = nn.ModuleList(nn.Embedding(1, embed_dim) for _ in range(4))
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.mask_downscaling = nn.Sequential(ConvParams(mask_in_chans // 4, 1, 2, 2), NormParams(mask_in_chans // 4, 1e-6), nn.GELU(),
                                              ConvParams(mask_in_chans, mask_in_chans // 4, 2, 2), NormParams(mask_in_chans, 1e-6), nn.GELU(),
                                              ConvParams(embed_dim, mask_in_chans, 1, 1))
        self.no_mask_embed = nn.Embedding(1, embed_dim)
        self._pe = None

    def dense_pe_tokens(self, dtype):
        key = (self.pe_layer.positional_encoding_gaussian_matrix.data_ptr(), dtype)
        if self._pe is None or self._pe[0] != key:
            self._pe = (key, self.pe_layer.dense_tokens(*self.image_embedding_size).to(dtype).contiguous())
        return self._pe[1]


# ------------------------------------------------------------------------------------------------ memory
class RoPEAttention(Attention):
    """reference sam2.py:1484-1548: 1 head, axial complex RoPE on q and the first (Nk - exclude) keys (table tiled)."""

    def __init__(self, *a, rope_theta=10000.0, rope_k_repeat=False, feat_sizes=(32, 32), **k):
        super().__init__(*a, **k)
        self.rope_theta, self.rope_k_repeat = rope_theta, rope_k_repeat
        self._tab = {}

    def table(self, nq, device):
        key = (nq, str(device))
        if key not in self._tab:
            dim = self.internal_dim // self.num_heads
            side = int(math.sqrt(nq))
            fr = 1.0 / (self.rope_theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
            t = torch.arange(side * side, dtype=torch.float32)
            ang = torch.cat([torch.outer(t % side, fr), torch.outer(torch.div(t, side, rounding_mode="floor"), fr)], dim=-1)
            self._tab[key] = (ang.cos().contiguous().to(device), ang.sin().contiguous().to(device))
        return self._tab[key]

    def forward(self, q, k, v, nq, nk, num_k_exclude_rope=0, residual=None):
        """single sequence: q [nq, C], k/v [nk, Ckv]."""
        assert self.num_heads == 1
        qp, kp, vp = self.q_proj(q), self.k_proj(k), self.v_proj(v)
        cos, sin = self.table(nq, q.device)
        ops.rope_axial_(qp, cos, sin, nq)
        ops.rope_axial_(kp, cos, sin, nk - num_k_exclude_rope)
        D = self.internal_dim
        o = ops.attn_varlen(qp.view(nq, 1, D), kp.view(nk, 1, D), vp.view(nk, 1, D), _cu(1, nq, q.device), _cu(1, nk, q.device), nq, D ** -0.5)
        return self.out_proj(o.view(nq, D), residual=residual)


class MemoryAttentionLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward, kv_in_dim):
        super().__init__()
        self.self_attn = RoPEAttention(d_model, 1)
        self.cross_attn_image = RoPEAttention(d_model, 1, rope_k_repeat=True, kv_in_dim=kv_in_dim)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = NormParams(d_model, 1e-5), NormParams(d_model, 1e-5), NormParams(d_model, 1e-5)

    def forward(self, x, mem_k, mem_v, nq, nk, n_excl):
        t = self.norm1(x)
        x = self.self_attn(t, t, t, nq, nq, residual=x)
        t = self.norm2(x)
        x = self.cross_attn_image(t, mem_k, mem_v, nq, nk, num_k_exclude_rope=n_excl, residual=x)
        t = self.norm3(x)
        return self.linear2(self.linear1(t, act="relu"), residual=x)


class MemoryAttention(nn.Module):
    """reference sam2.py:533-600 (single object / single sequence per call; dropout inactive in eval).

    Inference path, MI355X form (round 3; the layers' own forward above stays as the plain restatement and as the path for LoRA-wrapped projections):
      * the key projection of ALL layers is one product over the bank, [nk, 64] x [64, L*256], + one axial-RoPE pass (the bank does not depend on the layer input);
      * the bank is never projected to values: the cross-attention kernel accumulates softmax(S) memory in the 64-wide memory space (csrc/memattn.hip) and the value
        projection is applied to the 4096 query rows, folded with the output projection: out = PM (Wo Wv)^T + (Wo bv + bo);
      * q / k / v of the self-attention are one product ([256 -> 768]) and one RoPE pass over the q | k columns."""

    def __init__(self, d_model, num_layers, dim_feedforward, kv_in_dim):
        super().__init__()
        self.d_model = d_model
        self.layers = nn.ModuleList(MemoryAttentionLayer(d_model, dim_feedforward, kv_in_dim) for _ in range(num_layers))
        self.norm = NormParams(d_model, 1e-5)
        self._pk = None

    def _fusable(self):
        return not any(hasattr(m_, "lora_A") for l in self.layers for m_ in (l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj, l.self_attn.out_proj,
                                                                             l.cross_attn_image.q_proj, l.cross_attn_image.k_proj, l.cross_attn_image.v_proj,
                                                                             l.cross_attn_image.out_proj))

    def _packs(self):
        """Derived operands, rebuilt only when a source weight changes (frozen at inference: built once).  Elementwise torch ops + this library's own GEMM."""
        from .qwen2_5_vl import _versions
        srcs = []
        for l in self.layers:
            for a in (l.self_attn, l.cross_attn_image):
                for m_ in (a.q_proj, a.k_proj, a.v_proj, a.out_proj):
                    srcs += [m_.weight, m_.bias]
        ver = _versions(*srcs)
        if self._pk is not None and self._pk[0] == ver:
            return self._pk[1]
        with torch.no_grad():
            pk = {"wk_all": torch.cat([l.cross_attn_image.k_proj.weight for l in self.layers], 0).contiguous(),
                  "bk_all": torch.cat([l.cross_attn_image.k_proj.bias for l in self.layers], 0).contiguous(), "layers": []}
            for l in self.layers:
                sa, ca = l.self_attn, l.cross_attn_image
                wov = ops.gemm(ca.out_proj.weight, ops.transpose(ca.v_proj.weight.contiguous()), out_dtype=torch.float32).to(ca.out_proj.weight.dtype)   # Wo Wv [256, 64]
                bov = ops.gemm(ca.v_proj.bias.view(1, -1).contiguous(), ca.out_proj.weight, bias=ca.out_proj.bias, out_dtype=torch.float32).view(-1).to(ca.out_proj.weight.dtype)
                pk["layers"].append({"wqkv": torch.cat([sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight], 0).contiguous(),
                                     "bqkv": torch.cat([sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias], 0).contiguous(), "wov": wov.contiguous(), "bov": bov.contiguous()})
        self._pk = (ver, pk)
        return pk

    def _tables(self, nq, reps, device):
        """cos / sin of the axial RoPE tiled `reps` times along the columns (one pass rotates `reps` 256-wide column groups of a fused projection)."""
        ra = self.layers[0].self_attn
        key = ("tiled", nq, reps, str(device))
        if key not in ra._tab:
            cos, sin = ra.table(nq, device)
            ra._tab[key] = (cos.repeat(1, reps).contiguous(), sin.repeat(1, reps).contiguous())
        return ra._tab[key]

    def forward(self, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens):
        nq, nk = curr.shape[0], memory.shape[0]
        x = ops.add_bcast(curr, curr_pos, alpha=0.1)
        mem_k = ops.add(memory, memory_pos)
        if _ag() or not self._fusable() or self.d_model != 256 or memory.shape[1] != 64:
            for layer in self.layers:
                x = layer(x, mem_k, memory, nq, nk, num_obj_ptr_tokens)
            return self.norm(x)
        pk = self._packs()
        L, D = len(self.layers), self.d_model
        cosL, sinL = self._tables(nq, L, x.device)
        k_all = ops.gemm(mem_k, pk["wk_all"], pk["bk_all"])                           # keys of every layer: [nk, L * 256]
        ops.rope_axial_(k_all, cosL, sinL, nk - num_obj_ptr_tokens)
        cos2, sin2 = self._tables(nq, 2, x.device)
        cos1, sin1 = self.layers[0].self_attn.table(nq, x.device)
        cu = _cu(1, nq, x.device)
        memory = memory.contiguous()
        if _ROWCHAIN:
            # the row-wise steps between the attention kernels as ONE launch each (csrc/memlayer.hip): [norm1 -> qkv -> RoPE], [out_proj + residual -> norm2 -> q_proj ->
            # RoPE], [merge of the cross-attention slices -> Wo Wv + residual -> norm3]
            rope = (cos1, sin1)
            l0 = self.layers[0]
            qkv = ops.memlayer_rows(x, (l0.norm1.weight, l0.norm1.bias), l0.norm1.eps, w2=pk["layers"][0]["wqkv"], b2=pk["layers"][0]["bqkv"], rope=rope, rope_cols=2 * D)[2]
            for li, layer in enumerate(self.layers):
                lp, sa, ca = pk["layers"][li], layer.self_attn, layer.cross_attn_image
                q3, k3, v3 = (qkv[:, i * D:(i + 1) * D].unflatten(1, (1, D)) for i in range(3))
                o = ops.attn_varlen(q3, k3, v3, cu, cu, nq, D ** -0.5)
                x, _, qp = ops.memlayer_rows(x, (layer.norm2.weight, layer.norm2.bias), layer.norm2.eps, a=o.view(nq, D), w1=sa.out_proj.weight, b1=sa.out_proj.bias,
                                             w2=ca.q_proj.weight, b2=ca.q_proj.bias, rope=rope, rope_cols=D)
                parts = ops.memattn_cross(qp, k_all[:, li * D:(li + 1) * D], memory, D ** -0.5, partials=True)
                x, t, _ = ops.memlayer_rows(x, (layer.norm3.weight, layer.norm3.bias), layer.norm3.eps, partials=parts, w1=lp["wov"], b1=lp["bov"], want_t=True)
                x = layer.linear2(layer.linear1(t, act="relu"), residual=x)
                if li + 1 < L:
                    nx = self.layers[li + 1]
                    qkv = ops.memlayer_rows(x, (nx.norm1.weight, nx.norm1.bias), nx.norm1.eps, w2=pk["layers"][li + 1]["wqkv"], b2=pk["layers"][li + 1]["bqkv"], rope=rope,
                                            rope_cols=2 * D)[2]
            return self.norm(x)
        for li, layer in enumerate(self.layers):
            lp = pk["layers"][li]
            t = layer.norm1(x)
            qkv = ops.gemm(t, lp["wqkv"], lp["bqkv"])                                 # [nq, 768] = q | k | v
            ops.rope_axial_(qkv[:, :2 * D], cos2, sin2, nq)
            q3, k3, v3 = (qkv[:, i * D:(i + 1) * D].unflatten(1, (1, D)) for i in range(3))
            o = ops.attn_varlen(q3, k3, v3, cu, cu, nq, D ** -0.5)
            x = layer.self_attn.out_proj(o.view(nq, D), residual=x)
            t = layer.norm2(x)
            qp = layer.cross_attn_image.q_proj(t)
            ops.rope_axial_(qp, cos1, sin1, nq)
            pm = ops.memattn_cross(qp, k_all[:, li * D:(li + 1) * D], memory, D ** -0.5)   # softmax(S) memory, [nq, 64]
            x = ops.gemm(pm, lp["wov"], lp["bov"], residual=x)                        # value + output projection on the query rows
            t = layer.norm3(x)
            x = layer.linear2(layer.linear1(t, act="relu"), residual=x)
        return self.norm(x)


class CXBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = ConvParams(dim, 1, 7, 7)
        self.norm = NormParams(dim, 1e-6)
        self.pwconv1 = Linear(dim, 4 * dim)
        self.pwconv2 = Linear(4 * dim, dim)
        self.g_weight = nn.Parameter(1e-6 * torch.ones(dim))

    def forward(self, x, Fn, H, W):
        z = ops.dwconv7x7(x, self.dwconv.weight, self.dwconv.bias, Fn, H, W)
        z = self.norm(z)
        z = self.pwconv1(z, act="gelu")
        return ops.gemm(z, self.pwconv2.weight, self.pwconv2.bias, residual=x, colscale=self.g_weight)


class Fuser(nn.Module):
    def __init__(self, dim, num_layers):
        super().__init__()
        self.proj = nn.Identity()
        self.layers = nn.ModuleList(CXBlock(dim) for _ in range(num_layers))


class MaskDownSampler(nn.Module):
    def __init__(self, embed_dim, total_stride=16):
        super().__init__()
        n = int(math.log2(total_stride))
        self.encoder = nn.Sequential()
        cin = 1
        for _ in range(n):
            cout = cin * 4
            self.encoder.append(ConvParams(cout, cin, 3, 3))
            self.encoder.append(NormParams(cout, 1e-6))
            self.encoder.append(nn.GELU())
            cin = cout
        self.encoder.append(ConvParams(embed_dim, cin, 1, 1))
        self.n_down = n


class MemoryEncoder(nn.Module):
    """reference sam2.py:724-767"""

    def __init__(self, out_dim, in_dim, fuser_layers=2, total_stride=16):
        super().__init__()
        self.mask_downsampler = MaskDownSampler(in_dim, total_stride)
        self.pix_feat_proj = ConvParams(in_dim, in_dim, 1, 1)
        self.fuser = Fuser(in_dim, fuser_layers)
        self.out_proj = ConvParams(out_dim, in_dim, 1, 1)
        self.out_dim = out_dim
        self._pos = {}

    def forward(self, pix_tokens, mask_f32, Fn, S, sig_scale, sig_bias):
        """pix_tokens [Fn*h*w, C] raster; mask_f32 [Fn, S, S] raw high-res logits (sigmoid*scale+bias fused into the first conv)."""
        enc = self.mask_downsampler.encoder
        x, H = mask_f32.contiguous(), S
        for i in range(self.mask_downsampler.n_down):
            conv, norm = enc[3 * i], enc[3 * i + 1]
            ss, sb = (sig_scale, sig_bias) if i == 0 else (0.0, 0.0)
            if not _ag() and ops.conv3x3s2_ln_gelu_ok(x, conv.weight):   # the two narrow stages: convolution + LayerNorm2d + GELU in one launch
                x = ops.conv3x3s2_ln_gelu(x, conv.weight, conv.bias, norm.weight, norm.bias, norm.eps, Fn, H, H, ss, sb)
                H //= 2
                continue
            x = ops.conv3x3s2(x, conv.weight, conv.bias, Fn, H, H, ss, sb)
            H //= 2
            x = norm(x, act="gelu")   # incl. the 4-channel stage (narrow-row LayerNorm kernel)
        last = enc[3 * self.mask_downsampler.n_down]
        m = ops.gemm(x, last.as_linear(), last.bias)
        y = ops.gemm(pix_tokens, self.pix_feat_proj.as_linear(), self.pix_feat_proj.bias, residual=m)
        for layer in self.fuser.layers:
            y = layer(y, Fn, H, H)
        y = ops.gemm(y, self.out_proj.as_linear(), self.out_proj.bias)
        key = (H, str(y.device))
        if key not in self._pos:
            self._pos[key] = position_embedding_sine(self.out_dim, H, H, y.device).to(y.dtype)
        return y, self._pos[key]


# ------------------------------------------------------------------------------------------------ the model
class SAM2VideoPredictor(nn.Module):
    """Parameter container + frame-level arithmetic of reference SAM2Base / SAM2VideoPredictor (sam2.py:2348-4132)
    with the hyper-parameters the RGA3 wrapper hard-codes (sam2.py:97-136)."""

    def __init__(self, image_size=1024, embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43),
                 window_spec=(8, 4, 16, 8), pos_bkg=(7, 7), d_model=256, mem_dim=64, num_maskmem=7, memattn_layers=4, memattn_ff=2048,
                 max_obj_ptrs_in_encoder=16, backbone_stride=16, dec_mlp_dim=2048, iou_head_hidden_dim=256):
        super().__init__()
        self.image_size, self.hidden_dim, self.mem_dim, self.num_maskmem = image_size, d_model, mem_dim, num_maskmem
        self.backbone_stride, self.max_obj_ptrs_in_encoder = backbone_stride, max_obj_ptrs_in_encoder
        self.sigmoid_scale_for_mem_enc, self.sigmoid_bias_for_mem_enc = 20.0, -10.0
        trunk = Hiera(embed_dim, num_heads, stages, global_att_blocks, window_spec, pos_bkg)
        self.image_encoder = ImageEncoder(trunk, FpnNeck(d_model, trunk.channel_list), scalp=1)
        self.mask_downsample = ConvParams(1, 1, 4, 4)  # present in checkpoints (sam2.py:2434); unused on the language path
        self.memory_attention = MemoryAttention(d_model, memattn_layers, memattn_ff, mem_dim)
        self.memory_encoder = MemoryEncoder(mem_dim, d_model)
        self.maskmem_tpos_enc = nn.Parameter(torch.zeros(num_maskmem, 1, 1, mem_dim))
        self.no_mem_embed = nn.Parameter(torch.zeros(1, 1, d_model))
        self.no_mem_pos_enc = nn.Parameter(torch.zeros(1, 1, d_model))
        self.no_obj_ptr = nn.Parameter(torch.zeros(1, d_model))
        for p in (self.maskmem_tpos_enc, self.no_mem_embed, self.no_mem_pos_enc, self.no_obj_ptr):
            nn.init.trunc_normal_(p, std=0.02)
        s = image_size // backbone_stride
        self.sam_image_embedding_size = s
        self.sam_prompt_encoder = PromptEncoder(d_model, (s, s), (image_size, image_size))
        self.sam_mask_decoder = MaskDecoder(d_model, 2, 8, dec_mlp_dim, iou_head_hidden_dim)
        self.obj_ptr_proj = MLP(d_model, d_model, d_model, 3)

    @property
    def device(self):
        return self.no_mem_embed.device

    @property
    def dtype(self):
        return self.no_mem_embed.dtype

    # -- image encoder --------------------------------------------------------------------------------------
    def encode_frozen(self, img):
        """The FROZEN part of forward_image: Hiera trunk + FPN neck (reference qwen_2_5_vl_sam2.py:121 freezes the grounding encoder; only sam_mask_decoder trains).
        It depends on nothing the optimizer updates, so a trainer may run it for the NEXT sample ahead of time (UniGRModel.prefetch_sam)."""
        Fn = img.shape[0]
        with torch.no_grad():
            stages = self.image_encoder.trunk(img.to(self.dtype))
            return self.image_encoder.neck(stages, Fn)[: len(stages) - self.image_encoder.scalp]

    def forward_image(self, img, frozen=None):
        """reference sam2.py:2790-2802 (+ FPN, scalp). Returns dict of raster token maps.  frozen: the result of encode_frozen(img) when it was computed ahead of time."""
        Fn = img.shape[0]
        lv = frozen if frozen is not None else self.encode_frozen(img)
        (f0, H0, W0), (f1, H1, W1), (f2, H2, W2) = lv
        dec = self.sam_mask_decoder   # conv_s0 / conv_s1 belong to the (trainable) mask decoder: they record autograd when enabled

        def conv1x1(conv, x):
            if _ag():
                return AG.linear(x, conv.weight.reshape(conv.weight.shape[0], -1), conv.bias)
            return ops.gemm(x, conv.as_linear(), conv.bias)

        return {"feat_s0": conv1x1(dec.conv_s0, f0), "feat_s1": conv1x1(dec.conv_s1, f1),
                "feat": f2, "hw": (H2, W2), "n": Fn, "pos": self.image_encoder.neck.pos(H2, W2, f2.device, f2.dtype)}

    # -- SAM heads (language path: no clicks, no mask prompt) ---------------------------------------------
    def forward_sam_heads(self, pix_tokens, feats, language_embd, frame_slice=None):
        """reference sam2.py:3262-3431. pix_tokens [B*h*w, C] (memory-conditioned or + no_mem_embed)."""
        h, w = feats["hw"]
        B = pix_tokens.shape[0] // (h * w)
        pe = self.sam_prompt_encoder
        sparse = pe.not_a_point_embed.weight.expand(2, -1)[None].expand(B, -1, -1)
        if language_embd is not None:
            sparse = torch.cat([sparse, language_embd.to(sparse.dtype)], dim=1)
        src = ops.add_bcast(pix_tokens, pe.no_mask_embed.weight)  # dense prompt = no_mask_embed everywhere
        s0, s1 = feats["feat_s0"], feats["feat_s1"]
        if frame_slice is not None:
            a, b = frame_slice
            s0, s1 = s0[a * 16 * h * w: b * 16 * h * w], s1[a * 4 * h * w: b * 4 * h * w]
        masks, iou, toks, obj = self.sam_mask_decoder(src, pe.dense_pe_tokens(src.dtype), sparse, s0, s1, B, h, w, const_sparse=language_embd is None)
        proj = self.obj_ptr_proj
        if (not _ag() and proj.num_layers == 3 and proj.act == "relu" and not any(hasattr(l, "lora_A") for l in proj.layers) and toks.shape[2] % 8 == 0
                and toks.shape[2] <= 512 and iou.dtype == torch.bfloat16):
            # argmax over the multimask IoUs, chosen token -> obj_ptr_proj, object gating: one launch (csrc/dechead.hip)
            best, sel, sel64, obj_ptr = ops.sam_select_objptr(iou.contiguous(), obj.contiguous(), toks, tuple(t for l in proj.layers for t in (l.weight, l.bias)),
                                                       self.no_obj_ptr.view(-1))
            low = masks.reshape(B * 4, 4 * h, 4 * w)[sel64].unsqueeze(1)                              # chosen candidate, f32
            high = ops.bilinear(masks.view(B * 4, 4 * h, 4 * w), (self.image_size, self.image_size), sel).unsqueeze(1)
            # ("ious" is converted when somebody reads it: nothing on the tracking path does, and a launch there is 4.4 us of a 1.3-ms frame)
            return _HeadsOut({"low_res_multimasks": masks[:, 1:], "low_res_masks": low, "high_res_masks": high, "obj_ptr": obj_ptr,
                              "object_score_logits": obj, "best_iou_inds": best}, iou)
        ious = iou[:, 1:].float()
        best = torch.argmax(ious, dim=-1)
        bi = torch.arange(B, device=best.device)
        sel = (bi * 4 + 1 + best).to(torch.int32)
        low = masks.reshape(B * 4, 4 * h, 4 * w)[sel.long()].unsqueeze(1)                               # chosen candidate, f32
        if _ag():
            high = AG.BilinearFn.apply(masks.reshape(B * 4, 4 * h, 4 * w), (self.image_size, self.image_size), sel).unsqueeze(1)
            if _DEBUG is not None:
                high.retain_grad()
                _DEBUG["high"] = high
        else:
            high = ops.bilinear(masks.view(B * 4, 4 * h, 4 * w), (self.image_size, self.image_size), sel).unsqueeze(1)
        tok = toks[:, 1:][bi, best]
        obj_ptr = self.obj_ptr_proj(tok.contiguous())
        lam = (obj > 0).to(obj_ptr.dtype)
        obj_ptr = lam * obj_ptr + (1 - lam) * self.no_obj_ptr
        return {"low_res_multimasks": masks[:, 1:], "ious": ious, "low_res_masks": low, "high_res_masks": high, "obj_ptr": obj_ptr,
                "object_score_logits": obj, "best_iou_inds": best}

    # -- memory ----------------------------------------------------------------------------------------------
    def encode_new_memory(self, feats, frame, high_res_masks):
        """reference sam2.py:2991-3029 for one frame: (maskmem tokens [h*w, mem_dim], pos tokens)."""
        h, w = feats["hw"]
        pix = feats["feat"][frame * h * w:(frame + 1) * h * w]
        return self.memory_encoder(pix, high_res_masks.reshape(1, self.image_size, self.image_size).float(), 1, self.image_size,
                                   self.sigmoid_scale_for_mem_enc, self.sigmoid_bias_for_mem_enc)


class SAM2(nn.Module):
    """Drop-in for the reference wrapper (model/sam2.py:87-446)."""

    def __init__(self, ckpt_path: str = None, **tiny_overrides):
        super().__init__()
        self.sam2_model = SAM2VideoPredictor(**tiny_overrides)
        self.hidden_dim = self.sam2_model.hidden_dim
        self.img_mean, self.img_std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
        if ckpt_path is not None:
            load_sam2_checkpoint(self.sam2_model, ckpt_path)

    # ---- training path: frames independent (reference :412-433, :343-375)
    def get_sam2_embeddings_train(self, images, expand_size=1, frozen=None):
        assert expand_size == 1, "num_objs == 1 on the RGA3 path (model/qwen_2_5_vl_sam2.py:263)"
        return self.sam2_model.forward_image(images, frozen=frozen)

    def inject_language_embd_train(self, sam_states, language_embd, nf_nobj=None):
        m = self.sam2_model
        pix = ops.add_bcast(sam_states["feat"], m.no_mem_embed.view(1, -1))
        o = m.forward_sam_heads(pix, sam_states, language_embd)
        return o["low_res_masks"], o["high_res_masks"]

    # ---- inference path (reference :406-410, :378-404)
    def get_sam2_embeddings(self, images):
        return VideoSession(self.sam2_model, images)

    get_sam2_embeddings_inference = get_sam2_embeddings

    def language_embd_inference(self, session, language_embd):
        """Prompt every object on every frame, then propagate: every frame is an initial conditioning frame, so the masks are the consolidated low-res predictions
        upsampled to image_size (reference :378-404, :3749-3769).  language_embd[t] is [n_obj, C] ([C] / [1, C] for one object).  Returns [T * n_obj, 1, S, S] f32:
        the reference concatenates the per-frame yields [n_obj, 1, S, S] of propagate_in_video on dim 0 (:399-403), i.e. FRAME-major (frame 0 obj 0, frame 0 obj 1,
        frame 1 obj 0, ...) -- [T, 1, S, S] for the single object of every RGA3 caller; pinned at n_obj = 2 by tests/golden/sam2_multiobj.npz."""
        T = len(language_embd)
        # language_embd[t]: a tensor [n_obj, C] / [C] or a list of n_obj tensors (the reference indexes language_embd[frame_idx][obj_idx], :385-388)
        embs = [(torch.stack([x.reshape(-1) for x in e]) if isinstance(e, (list, tuple)) else e.reshape(-1, e.shape[-1])) for e in language_embd]
        n_obj = embs[0].shape[0]
        outs = []
        for o in range(n_obj):   # objects do not interact (non_overlap_masks / non_overlap_masks_for_mem_enc / clear_non_cond_mem_* are False: reference :2392, :3512-3517)
            sess = session if o == 0 else VideoSession(self.sam2_model, session.images, feats=session._ensure_feats())
            for t in range(T):
                sess.add_language_embd(t, embs[t][o].reshape(1, 1, -1), use_graph=True)
            outs.append(torch.cat([mk for _, mk in sess.propagate()], dim=0))     # [T, 1, S, S]
        if n_obj == 1:
            return outs[0]
        return torch.stack(outs, dim=1).reshape(T * n_obj, 1, *outs[0].shape[-2:])

    def get_sam2_embeddings_multi(self, images, n_obj: int):
        """A clip tracked for n_obj objects on shared image features (the reference's init_state + add_language_embd with several obj_ids, :3771-3975)."""
        return MultiObjectSession(self.sam2_model, images, n_obj)

    def forward(self, batch):
        raise NotImplementedError


def _graph_cache(m, session=None):
    """The captured frame graphs of model m.  They read the weights (and the packs built from them) through their pointers, so any
    in-place parameter update since the capture -- seen as a bumped tensor version -- drops them all; the next frame re-captures.
    The check walks ~900 parameters, so a session does it once (weights do not change inside a session)."""
    cache = m.__dict__.setdefault("_frame_graphs", {})
    if session is None or not getattr(session, "_graphs_checked", False):
        sig = tuple((p.data_ptr(), p._version) for p in m.parameters())
        if cache.get("sig") != sig:
            cache.clear()
            cache["sig"] = sig
        if session is not None:
            session._graphs_checked = True
    slot = getattr(session, "slot", 0) if session is not None else 0
    if ("pool", slot) not in cache:
        cache[("pool", slot)] = torch.cuda.graph_pool_handle()       # the graphs of one slot never run concurrently: one private pool per slot
        # ... and one CAPTURE stream + one scratch scope per slot: ops.py keys its scratch (stream-K slabs / flags, the memory-attention partial sums, the TN
        # counters) by (device, current stream), and the pointers a capture sees are baked into the graph -- slots whose graphs replay concurrently
        # (MultiObjectSession) must not share them.  A capture stream of its own is not enough: torch.cuda.Stream() cycles through a pool of 32,
        # so some later eager stream IS a slot's capture stream and would share that graph's scratch while the graph replays elsewhere -- warm-up and capture
        # therefore run inside ops.workspace_scope(("sam2", id(m), slot)): the baked scratch is keyed by the slot, and no eager launch can ever see it.
        cache[("stream", slot)] = torch.cuda.Stream()
        cache[("scope", slot)] = ("sam2-frames", id(m), slot)
    return cache


class VideoSession:
    """Single-object video state (reference SAM2VideoPredictor.init_state / add_language_embd / propagate_in_video,
    sam2.py:3771-4132).  Image features are computed once per frame and kept; the memory encoder runs only for frames
    whose memory can be read by a later frame."""

    _uids = 0

    def __init__(self, model: SAM2VideoPredictor, images, feats=None, chunk=8, slot: int = 0):
        self.m, self.images = model, images
        self.slot = int(slot)     # sessions that replay their frame graphs CONCURRENTLY (MultiObjectSession: one stream per object) own separate graphs, static buffers and pools
        self.num_frames = images.shape[0]
        self.cond: Dict[int, dict] = {}
        self.non_cond: Dict[int, dict] = {}
        self.temp_cond: Dict[int, dict] = {}
        self.counts = {"enc": 0, "memattn": 0, "memenc": 0, "dec": 0}
        self.feats = feats
        self.chunk = chunk
        VideoSession._uids += 1
        self.uid = VideoSession._uids

    def _ensure_feats(self):
        if self.feats is None:
            parts = []
            with torch.no_grad():
                for a in range(0, self.num_frames, self.chunk):
                    parts.append(self.m.forward_image(self.images[a:a + self.chunk]))
                    self.counts["enc"] += min(self.chunk, self.num_frames - a)
            self.feats = {k: torch.cat([p[k] for p in parts]) for k in ("feat_s0", "feat_s1", "feat")}
            self.feats.update(hw=parts[0]["hw"], n=self.num_frames, pos=parts[0]["pos"])
        return self.feats

    def _frame_tokens(self, t):
        f = self._ensure_feats()
        h, w = f["hw"]
        return f["feat"][t * h * w:(t + 1) * h * w]

    def add_language_embd(self, frame_idx, language_embd, use_graph: bool = False):
        """Language prompt on one frame (reference add_language_embd, sam2.py:3824-3975): mask decoder on the frame's features without memory.
        use_graph: the step (≈100 short launches, the same shapes for every frame) is captured once per model and replayed."""
        f = self._ensure_feats()
        if use_graph and not _ag():
            o = self._graph_prompt(frame_idx, language_embd)
        else:
            pix = ops.add_bcast(self._frame_tokens(frame_idx), self.m.no_mem_embed.view(1, -1))
            o = self.m.forward_sam_heads(pix, f, language_embd, frame_slice=(frame_idx, frame_idx + 1))
        self.counts["dec"] += 1
        self.temp_cond[frame_idx] = {"pred_masks": o["low_res_masks"], "obj_ptr": o["obj_ptr"], "best_iou_inds": o["best_iou_inds"]}
        return o["low_res_masks"]

    def _graph_prompt(self, t, language_embd):
        m, f = self.m, self._ensure_feats()
        h, w = f["hw"]
        hw, dev = h * w, f["feat"].device
        cache = _graph_cache(m, self)
        key = ("prompt", hw, tuple(language_embd.shape), str(dev), f["feat"].dtype, self.slot)
        ent = cache.get(key)
        fresh = ent is None
        if fresh:
            ent = cache[key] = {"G": {"tok": torch.empty_like(f["feat"][:hw]), "s0": torch.empty_like(f["feat_s0"][:16 * hw]),
                                      "s1": torch.empty_like(f["feat_s1"][:4 * hw]), "emb": torch.empty_like(language_embd), "pos": torch.empty_like(f["pos"])}}
        G = ent["G"]
        G["tok"].copy_(f["feat"][t * hw:(t + 1) * hw])
        G["s0"].copy_(f["feat_s0"][t * 16 * hw:(t + 1) * 16 * hw])
        G["s1"].copy_(f["feat_s1"][t * 4 * hw:(t + 1) * 4 * hw])
        G["emb"].copy_(language_embd)
        G["pos"].copy_(f["pos"])     # everything the captured launches read lives in G (the graph outlives this session)
        feats1 = {"feat_s0": G["s0"], "feat_s1": G["s1"], "feat": G["tok"], "hw": (h, w), "n": 1, "pos": G["pos"]}

        def body():
            pix = ops.add_bcast(G["tok"], m.no_mem_embed.view(1, -1))
            o = m.forward_sam_heads(pix, feats1, G["emb"], frame_slice=(0, 1))
            return o["low_res_masks"], o["obj_ptr"], o["best_iou_inds"]

        if fresh:
            side = cache[("stream", self.slot)]
            side.wait_stream(torch.cuda.current_stream())
            with ops.workspace_scope(cache[("scope", self.slot)]):
                with torch.cuda.stream(side):
                    body()
                torch.cuda.current_stream().wait_stream(side)
                ent["graph"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ent["graph"], pool=cache[("pool", self.slot)], stream=side):
                    ent["outs"] = body()
        ent["graph"].replay()
        low, ptr, best = ent["outs"]
        return {"low_res_masks": low.clone(), "obj_ptr": ptr.clone(), "best_iou_inds": best.clone()}

    def _memory_for(self, t, low_res_masks):
        S = self.m.image_size
        high = ops.bilinear(low_res_masks.reshape(1, *low_res_masks.shape[-2:]).float().contiguous(), (S, S))
        mf, mp = self.m.encode_new_memory(self._ensure_feats(), t, high)
        self.counts["memenc"] += 1
        return mf, mp

    def _preflight(self, need_memory):
        for t, cur in sorted(self.temp_cond.items()):
            out = dict(cur)
            if need_memory:
                out["maskmem_features"], out["maskmem_pos_enc"] = self._memory_for(t, cur["pred_masks"])
            self.cond[t] = out
        self.temp_cond = {}

    def _memory_of(self, tt, out):
        """maskmem features / positions of an already-tracked frame, encoded on first use: frames whose memory no later frame of the SAME pass reads are left without
        one (propagate), and a later pass -- reverse tracking from a middle frame reads what the forward pass left on the frames after it -- encodes it then, from the
        same low-resolution mask through the same bilinear map, i.e. the same numbers the reference stored at tracking time (reference sam2.py:3215-3259, :3017-3029)."""
        if out.get("maskmem_features") is None:
            out["maskmem_features"], out["maskmem_pos_enc"] = self._memory_for(tt, out["pred_masks"])
        return out["maskmem_features"], out["maskmem_pos_enc"]

    def _conditioned_features(self, t, reverse=False):
        """reference sam2.py:2820-2989 (r = 1, no temporal enc on pointers).  Forward: memories of frames t - 6 .. t - 1 and pointers of the frames before t; reverse
        (track_in_reverse): frames t + 1 .. t + 6 and pointers of the frames after t -- whichever pass left them (:2866-2893, :2927-2944)."""
        m = self.m
        h, w = self._ensure_feats()["hw"]
        sgn = 1 if reverse else -1
        to_cat, to_cat_pos = [], []
        prevs = [(0, tt, o) for tt, o in self.cond.items()]
        for t_pos in range(1, m.num_maskmem):
            tt = t + sgn * (m.num_maskmem - t_pos)
            prevs.append((t_pos, tt, self.non_cond.get(tt, None)))
        for t_pos, tt, prev in prevs:
            if prev is None:
                continue
            mf, mp = self._memory_of(tt, prev)
            to_cat.append(mf)
            to_cat_pos.append(ops.add_bcast(mp, m.maskmem_tpos_enc[m.num_maskmem - t_pos - 1].view(1, -1)))
        ptrs = [o["obj_ptr"] for tt, o in self.cond.items() if (tt >= t if reverse else tt <= t)]
        for t_diff in range(1, min(self.num_frames, m.max_obj_ptrs_in_encoder)):
            tt = t + sgn * t_diff
            if tt < 0 or tt >= self.num_frames:
                break
            o = self.non_cond.get(tt, None)
            if o is not None:
                ptrs.append(o["obj_ptr"])
        n_ptr = 0
        if ptrs:
            op = torch.cat(ptrs, dim=0).reshape(-1, m.mem_dim)  # each 1 x C pointer -> C/mem_dim tokens (sam2.py:2952-2958)
            to_cat.append(op)
            to_cat_pos.append(torch.zeros_like(op))
            n_ptr = op.shape[0]
        memory, mpos = torch.cat(to_cat, dim=0).contiguous(), torch.cat(to_cat_pos, dim=0).contiguous()
        self.counts["memattn"] += 1
        return m.memory_attention(self._frame_tokens(t), self._ensure_feats()["pos"], memory, mpos, n_ptr)

    # -- frames as hipGraph replays ------------------------------------------------------------------------------------------
    # With one conditioning frame the bank of frame t depends only on d = t - start: min(d-1, 6) earlier non-conditioning memories, always in
    # the same slots (slot = age, so the memory positions are constants), and min(d-1, 15) + 1 pointers; from d = 16 on nothing changes
    # shape any more.  A frame is then a pure function of (its tokens, its two high-resolution feature maps, the bank) made of ~200 short
    # launches -- 2 % MFMA use, bounded by launch / ramp latency -- so each of the 16 bank states is captured ONCE per model into a graph
    # over static buffers and replayed for every later frame and stream: per frame the host issues the copies into the static inputs, one
    # graph launch and the copies out.  The captured launches read the weights through their pointers, so weight updates are seen.
    def _graph_frame(self, t, start):
        m, f = self.m, self._ensure_feats()
        h, w = f["hw"]
        hw, dev = h * w, f["feat"].device
        n_mem = m.num_maskmem
        max_pp = min(self.num_frames, m.max_obj_ptrs_in_encoder) - 1
        k = m.hidden_dim // m.mem_dim
        d = t - start
        n_nc, n_pp = min(d - 1, n_mem - 1), min(d - 1, max_pp)        # earlier non-conditioning memories / pointers in the bank
        cond = self.cond[start]
        cache = _graph_cache(m, self)
        key = (hw, n_nc, n_pp, str(dev), f["feat"].dtype, self.slot)
        ent = cache.get(key)
        fresh = ent is None
        if fresh:
            G = {"tok": torch.empty_like(f["feat"][:hw]), "s0": torch.empty_like(f["feat_s0"][:16 * hw]), "s1": torch.empty_like(f["feat_s1"][:4 * hw]),
                 "mem": torch.empty(((1 + n_nc) * hw + (1 + n_pp) * k, m.mem_dim), dtype=cond["maskmem_features"].dtype, device=dev),
                 "pos": torch.empty_like(f["pos"])}
            G["mem_pos"] = torch.empty_like(G["mem"])
            ent = cache[key] = {"G": G, "session": -1}
        G = ent["G"]
        p0 = (1 + n_nc) * hw                                             # first pointer row
        # Positions of the bank rows are a function of the bank STATE alone (the memory encoder's position map is one cached tensor per resolution, pointers carry
        # none): built once per captured state, not once per session -- it was 2 (1 + n_nc) eager launches on the first frame a session spent in each of its 17
        # states, 4 ms of a 42-ms stream.  Re-done only if the position map or the temporal table changes (identity + version).
        pos_src = cond["maskmem_pos_enc"]
        pos_key = (pos_src.data_ptr(), pos_src._version, m.maskmem_tpos_enc.data_ptr(), m.maskmem_tpos_enc._version)
        if ent.get("pos_key") != pos_key:
            G["mem_pos"][:hw].copy_(ops.add_bcast(pos_src, m.maskmem_tpos_enc[n_mem - 1].view(1, -1)))
            for i in range(n_nc):                                        # slot i holds frame t - n_nc + i, i.e. t_pos = n_mem - n_nc + i
                t_pos = n_mem - n_nc + i
                G["mem_pos"][(1 + i) * hw:(2 + i) * hw].copy_(ops.add_bcast(pos_src, m.maskmem_tpos_enc[n_mem - t_pos - 1].view(1, -1)))
            G["mem_pos"][p0:].zero_()
            ent["pos_key"] = pos_key
        moves = []
        if ent["session"] != self.uid:
            # constant part of the bank for this session: the image position map, the conditioning frame's memory and pointer -- they ride in the frame's one
            # copy launch below
            moves = [(G["pos"], f["pos"]), (G["mem"][:hw], cond["maskmem_features"]), (G["mem"][p0:p0 + k], cond["obj_ptr"].reshape(-1, m.mem_dim))]
            ent["session"] = self.uid
        feats1 = {"feat_s0": G["s0"], "feat_s1": G["s1"], "feat": G["tok"], "hw": (h, w), "n": 1, "pos": G["pos"]}
        S = m.image_size

        def body():
            pix = m.memory_attention(G["tok"], G["pos"], G["mem"], G["mem_pos"], (1 + n_pp) * k)
            o = m.forward_sam_heads(pix, feats1, None, frame_slice=(0, 1))
            mf, _ = m.encode_new_memory(feats1, 0, o["high_res_masks"])
            pm = o["low_res_masks"]
            # the frame's mask at video resolution IS high_res_masks here (both are the bilinear image of the chosen low-resolution mask at image_size; the reference
            # interpolates twice with the same mode, sam2.py:3388-3393 and :3761-3766): one launch instead of two
            mask = o["high_res_masks"].reshape(1, 1, S, S)
            return pm, o["obj_ptr"], mf, mask

        # the frame's inputs and the moving part of the bank into the graph's static buffers: one launch for all of them
        moves += [(G["tok"], f["feat"][t * hw:(t + 1) * hw]), (G["s0"], f["feat_s0"][t * 16 * hw:(t + 1) * 16 * hw]), (G["s1"], f["feat_s1"][t * 4 * hw:(t + 1) * 4 * hw])]
        moves += [(G["mem"][(1 + i) * hw:(2 + i) * hw], self.non_cond[t - n_nc + i]["maskmem_features"]) for i in range(n_nc)]
        moves += [(G["mem"][p0 + dd * k:p0 + (dd + 1) * k], self.non_cond[t - dd]["obj_ptr"].reshape(-1, m.mem_dim)) for dd in range(1, n_pp + 1)]
        ops.copy_many(moves)
        if fresh:
            side = cache[("stream", self.slot)]
            side.wait_stream(torch.cuda.current_stream())
            with ops.workspace_scope(cache[("scope", self.slot)]):
                with torch.cuda.stream(side):      # warm-up on the slot's capture stream: tuner picks, THIS slot's workspaces and lazily built tables exist before capture
                    body()
                torch.cuda.current_stream().wait_stream(side)
                ent["graph"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ent["graph"], pool=cache[("pool", self.slot)], stream=side):
                    ent["outs"] = body()
        ent["graph"].replay()
        outs = [torch.empty_like(o_) for o_ in ent["outs"]]      # the graph's result buffers are overwritten by the next replay
        ops.copy_many(list(zip(outs, ent["outs"])))
        return tuple(outs)

    def propagate(self, use_graph: bool = False, start_frame_idx=None, max_frame_num_to_track=None, reverse: bool = False):
        """reference propagate_in_video (sam2.py:4049-4132): (frame_idx, masks [1, 1, S, S] f32) in processing order -- by default every frame from the first
        conditioning frame on; `start_frame_idx`, `max_frame_num_to_track`, `reverse` as the reference defines them (reverse from frame 0 yields nothing, :4087-4090).
        use_graph: in a default forward stream with one conditioning frame every later frame runs as the replay of a captured hipGraph (one per bank state, kept on
        the model) -- same kernels, same results."""
        return list(self.propagate_iter(use_graph, start_frame_idx, max_frame_num_to_track, reverse))

    def propagate_iter(self, use_graph: bool = False, start_frame_idx=None, max_frame_num_to_track=None, reverse: bool = False):
        """propagate() one frame per next(): the reference's propagate_in_video IS a generator (:4049), and MultiObjectSession advances its objects frame by frame."""
        S = self.m.image_size
        all_cond = set(self.temp_cond) | set(self.cond)
        if not all_cond:
            raise RuntimeError("No points are provided; please add points first")     # the reference's message (:4066)
        default = start_frame_idx is None and max_frame_num_to_track is None and not reverse
        start = min(all_cond) if start_frame_idx is None else int(start_frame_idx)
        n_track = self.num_frames if max_frame_num_to_track is None else int(max_frame_num_to_track)
        if reverse:
            order = list(range(start, max(start - n_track, 0) - 1, -1)) if start > 0 else []
        else:
            order = list(range(start, min(start + n_track, self.num_frames - 1) + 1))
        need_memory = any(t not in all_cond for t in order)
        self._preflight(need_memory)
        graphed = use_graph and default and len(self.cond) == 1 and not _ag()
        for i, t in enumerate(order):
            if graphed and t not in self.cond:
                pm, ptr, mf, mask = self._graph_frame(t, start)
                self.counts["memattn"] += 1
                self.counts["dec"] += 1
                self.counts["memenc"] += 1
                self.non_cond[t] = {"pred_masks": pm, "obj_ptr": ptr, "maskmem_features": mf, "maskmem_pos_enc": self.cond[start]["maskmem_pos_enc"]}
                yield t, mask
                continue
            if t in self.cond:
                pm = self.cond[t]["pred_masks"]
            else:
                pix = self._conditioned_features(t, reverse)
                o = self.m.forward_sam_heads(pix, self._ensure_feats(), None, frame_slice=(t, t + 1))
                self.counts["dec"] += 1
                cur = {"pred_masks": o["low_res_masks"], "obj_ptr": o["obj_ptr"], "best_iou_inds": o["best_iou_inds"]}
                if any(tt not in self.cond for tt in order[i + 1:]):  # no later frame of this pass reads it otherwise (a later pass encodes it on first use: _memory_of)
                    self.counts["memenc"] += 1
                    cur["maskmem_features"], cur["maskmem_pos_enc"] = self.m.encode_new_memory(self._ensure_feats(), t, o["high_res_masks"])
                self.non_cond[t] = cur
                pm = cur["pred_masks"]
            yield t, ops.bilinear(pm.reshape(1, *pm.shape[-2:]).float().contiguous(), (S, S)).unsqueeze(0)


class MultiObjectSession:
    """n_obj objects tracked over one clip on SHARED image features (reference SAM2VideoPredictor with several obj_ids: add_language_embd per object, sam2.py:3824-3975;
    propagate_in_video runs every frame with batch_size = n_obj, :3977-4132, and yields video_res_masks [n_obj, 1, S, S]).  Nothing couples the batch entries in the
    reference (non_overlap_masks, non_overlap_masks_for_mem_enc, clear_non_cond_mem_* are False, :2392, :3512-3517), so an object is a VideoSession of its own: same
    kernels, same results as the single-object path, pinned against the reference's n_obj = 2 outputs (tests/golden/sam2_multiobj.npz).

    What the batch buys on this chip is CONCURRENCY, not wider kernels: a tracked frame is ~100 short, dependent launches that leave most CUs idle (the
    memory cross-attention excepted), so each object replays its captured frame graph on a stream of its own and the objects' frames overlap
    (`concurrent=True`; every object slot owns its graphs, static buffers and pool)."""

    def __init__(self, model: SAM2VideoPredictor, images, n_obj: int, feats=None, chunk=8):
        assert n_obj >= 1
        first = VideoSession(model, images, feats=feats, chunk=chunk, slot=0)
        f = first._ensure_feats()
        self.sessions = [first] + [VideoSession(model, images, feats=f, chunk=chunk, slot=o) for o in range(1, n_obj)]
        self.n_obj, self.num_frames = n_obj, first.num_frames
        self._streams = None

    @property
    def feats(self):
        return self.sessions[0].feats

    def add_language_embd(self, frame_idx, obj_idx, language_embd, use_graph: bool = False):
        return self.sessions[obj_idx].add_language_embd(frame_idx, language_embd, use_graph=use_graph)

    def propagate(self, use_graph: bool = False, concurrent: bool = True, **kw):
        """[(frame_idx, masks [n_obj, 1, S, S] f32)] in processing order (every object must have been prompted: the reference fills an un-prompted object's slot
        with NO_OBJ_SCORE masks, :3630-3747 -- not restated, no RGA3 caller does that)."""
        return list(self.propagate_iter(use_graph, concurrent, **kw))

    def propagate_iter(self, use_graph: bool = False, concurrent: bool = True, **kw):
        gens = [s.propagate_iter(use_graph=use_graph, **kw) for s in self.sessions]
        par = concurrent and self.n_obj > 1 and self.sessions[0].images.is_cuda
        if not par:
            while True:
                step = [next(g, None) for g in gens]
                if step[0] is None:
                    return
                yield step[0][0], torch.cat([m for _, m in step], dim=0)
        main = torch.cuda.current_stream()
        if self._streams is None:
            self._streams = [torch.cuda.Stream() for _ in range(self.n_obj)]
        for st in self._streams:
            st.wait_stream(main)          # the image features (and the prompts' outputs) were produced on the caller's stream
        while True:
            step = []
            for g, st in zip(gens, self._streams):
                with torch.cuda.stream(st):
                    step.append(next(g, None))
            if step[0] is None:
                for st in self._streams:
                    main.wait_stream(st)
                return
            for (_, m), st in zip(step, self._streams):
                main.wait_stream(st)
                m.record_stream(main)     # allocated on the object's stream, read by the concatenation on the caller's
            yield step[0][0], torch.cat([m for _, m in step], dim=0)


def load_sam2_checkpoint(model: SAM2VideoPredictor, path: str):
    """reference sam2.py:30-85: accept {'model': sd} / {'state_dict': sd} / sd, rename '.gamma' -> '.g_weight', strict."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck.get("state_dict", ck.get("model", ck)) if isinstance(ck, dict) else ck     # 'state_dict' before 'model', as the reference checks them (:47-52)
    sd = {k.replace(".gamma", ".g_weight"): v for k, v in sd.items()}                  # the reference's name_map (:70)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"SAM2 checkpoint mismatch: missing {missing[:5]}... unexpected {unexpected[:5]}...")
