"""Autograd nodes over the HIP ops for the trainable mask path (SAM2 mask decoder + text_hidden_fcs, reference
train_joint.py:237-251).  Each Function's forward/backward is a handful of kernel launches from rga3.hip.ops; torch only
records the graph.  (The decoder LLM uses one hand-written node per layer instead: rga3.model.qwen_train.)
"""
from __future__ import annotations

import torch

from . import ops


# W^T operands of the dX products: every weight whose input needs a gradient is noted in forward; the first backward that misses transposes ALL of them in one launch
# (ops.transpose_many) -- ~46 transposes of small matrices per training step were ~46 launches at the launch floor (profiles/r02_train_step_timeline.txt, 105-115 ms).
# Entries are checked against the weight's version, so an optimizer step in between simply misses.
_wt_pending, _wt_cache = {}, {}


def _transposed(w):
    key = (w.data_ptr(), w._version, tuple(w.shape))
    hit = _wt_cache.get(w.data_ptr())
    if hit is not None and hit[0] == key:
        return hit[1]
    todo = dict(_wt_pending)
    todo[w.data_ptr()] = w
    _wt_pending.clear()
    _wt_cache.clear()
    ws = list(todo.values())
    # every entry keeps its SOURCE tensor alive: while an entry exists no other tensor can be allocated at its address, so (address, version, shape) identifies the
    # weight (several of these are per-step temporaries -- ConvTranspose weights re-laid out, LoRA factors times their scale)
    for t, tt in zip(ws, ops.transpose_many(ws)):
        _wt_cache[t.data_ptr()] = ((t.data_ptr(), t._version, tuple(t.shape)), tt, t)
    return _wt_cache[w.data_ptr()][1]


class LinearFn(torch.autograd.Function):
    """y = act(a @ w^T + bias) (+ residual); act in {"none", "relu"}; out bf16 or f32."""

    @staticmethod
    def forward(ctx, a, w, bias, residual, act, out_f32):
        y = ops.gemm(a, w, bias, residual=residual, act=act, out_dtype=torch.float32 if out_f32 else torch.bfloat16)
        ctx.act, ctx.has_bias, ctx.has_res = act, bias is not None, residual is not None
        ctx.save_for_backward(a, w, y if act == "relu" else None)
        if a.requires_grad and w.dim() == 2 and w.is_contiguous() and w.element_size() == 2:
            _wt_pending[w.data_ptr()] = w.detach()
        return y

    @staticmethod
    def backward(ctx, dy):
        a, w, y = ctx.saved_tensors
        dy = dy.to(torch.bfloat16).contiguous()
        dres = dy if ctx.has_res else None
        if ctx.act == "relu":
            if ctx.has_res:
                raise NotImplementedError("relu + residual epilogue is not differentiated")
            dy = ops.act_bwd(y, dy, "relu")
        da = dw = db = None
        if ctx.needs_input_grad[0]:
            wd = w.detach()
            da = ops.gemm(dy, _transposed(wd) if (wd.dim() == 2 and wd.is_contiguous() and wd.element_size() == 2) else ops.transpose(wd))   # [M, K] = dy @ w
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn(dy, a.detach()).to(w.dtype)    # [N, K] = dy^T @ a, contraction over rows as they lie (per-pixel products: up to 64 K-slices)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy).to(torch.bfloat16)
        return da, dw, db, dres, None, None


def linear(a, w, bias=None, residual=None, act="none", out_f32=False):
    return LinearFn.apply(a, w, bias, residual, act, out_f32)


class DropoutFn(torch.autograd.Function):
    """Inverted dropout with the counter-hash mask of (seed, element index): backward re-applies the same mask from the seed."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed), None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        ctx.eps = eps
        ctx.save_for_backward(x, w)
        return ops.layernorm(x, w, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dw, db = ops.layernorm_bwd(x.contiguous(), w, dy.contiguous(), ctx.eps, want_param_grads=need_p)
        return dx, (dw.to(w.dtype) if need_p else None), (db.to(w.dtype) if need_p else None), None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(x.contiguous(), dy.contiguous(), "gelu")


class AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cu_q, cu_k, max_q, max_k, scale):
        o, lse = ops.attn_varlen(q, k, v, cu_q, cu_k, max_q, scale, False, return_lse=True, max_k=max_k)
        ctx.save_for_backward(q, k, v, o, lse, cu_q, cu_k)
        ctx.args = (max_q, max_k, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, cu_q, cu_k = ctx.saved_tensors
        max_q, max_k, scale = ctx.args
        dq, dk, dv = ops.attn_varlen_bwd(q, k, v, o, do.contiguous(), lse, cu_q, cu_k, max_q, max_k, scale, False)
        return dq, dk, dv, None, None, None, None, None


class AddBcastFn(torch.autograd.Function):
    """a + alpha * b with b broadcast over row blocks; gradient flows to a only (b is a constant positional table)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        return ops.add_bcast(a, b, alpha)

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class PixelShuffleFn(torch.autograd.Function):
    """ConvTranspose2d(k2,s2) tail: shuffle(g) + bias + add (add = frozen high-res feature, no grad)."""

    @staticmethod
    def forward(ctx, g, bias, add, F, H, W):
        ctx.dims = (F, H, W)
        return ops.pixel_shuffle2x(g, bias, add, F, H, W)

    @staticmethod
    def backward(ctx, dy):
        F, H, W = ctx.dims
        dy = dy.contiguous()
        dg = ops.pixel_shuffle2x_bwd(dy, F, H, W)
        db = ops.colsum(dy).to(torch.bfloat16) if ctx.needs_input_grad[1] else None
        return dg, db, (dy if ctx.needs_input_grad[2] else None), None, None, None


class BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, plane_idx):
        ctx.in_shape, ctx.plane_idx = tuple(x.shape), plane_idx
        return ops.bilinear(x.contiguous(), size, plane_idx)

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear_bwd(dy.contiguous().float(), ctx.in_shape, ctx.plane_idx), None, None


class MaskProductFn(torch.autograd.Function):
    """masks [B, 4, P] f32 = hyper [B, 4, C] x up [B * P, C]^T per frame (reference model/sam2.py:2142-2149), all frames in one launch each way."""

    @staticmethod
    def forward(ctx, hyper, up, P):
        hyper, up = hyper.contiguous(), up.contiguous()
        ctx.P = P
        ctx.save_for_backward(hyper, up)
        return ops.mask_product(hyper, up, P)

    @staticmethod
    def backward(ctx, dm):
        hyper, up = ctx.saved_tensors
        dhyper, dup = ops.mask_product_bwd(dm.float().contiguous(), hyper, up, ctx.P)
        return dhyper, dup, None


class MaskLossFn(torch.autograd.Function):
    """(sum_n mean-BCE_n, sum_n dice_n) over masks pred/target f32 [n, h, w] (reference qwen_2_5_vl_sam2.py:17-60)."""

    @staticmethod
    def forward(ctx, pred, target):
        pred, target = pred.contiguous(), target.float().contiguous()
        s = ops.bce_dice_sums(pred, target)
        hw = pred[0].numel()
        bce = (s[:, 0] / hw).sum()
        dice = (1 - (2 * s[:, 1] / 1000 + 1e-6) / (s[:, 2] / 1000 + s[:, 3] / 1000 + 1e-6)).sum()
        ctx.save_for_backward(pred, target, s)
        return bce, dice

    @staticmethod
    def backward(ctx, gb, gd):
        pred, target, s = ctx.saved_tensors
        return ops.bce_dice_grad(pred, target, s, gb, gd), None   # gb / gd stay on the device


class GatherRowsFn(torch.autograd.Function):
    """rows = x[idx] for UNIQUE idx (backward scatters into zeros)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.n = x.shape[0]
        ctx.save_for_backward(idx)
        return ops.gather_rows(x, idx)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros((ctx.n, dy.shape[1]), dtype=dy.dtype, device=dy.device)
        ops.scatter_rows_(dx, idx, dy.contiguous())
        return dx, None


class ScatterRowsFn(torch.autograd.Function):
    """out = zeros[n_rows]; out[idx] = src (unique idx); backward gathers."""

    @staticmethod
    def forward(ctx, src, idx, n_rows):
        ctx.save_for_backward(idx)
        out = torch.zeros((n_rows, src.shape[1]), dtype=src.dtype, device=src.device)
        ops.scatter_rows_(out, idx, src.contiguous())
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.gather_rows(dy.contiguous(), idx), None, None
