"""ORACLE — test infrastructure only.  The config-5 training step of the fp32 restatement with the FROZEN decoder contractions in e4m3, exactly as the
build runs them (rga3.model.qwen_train._fgemm): activations quantised per token, weights per output row (scale = amax / 448, e4m3 RNE: kernels_ref.
quant_fp8_rows_ref, bit-exact against rga3_quant_fp8_rows), products of e4m3 values summed in fp32, result rounded to bf16; backward dX through the same
scheme on (dY, W^T).  LoRA factors, norms, attention, lm_head and embeddings stay in the oracle's fp32.  With this the fp8 step of the HIP path is
compared with an ORACLE fp8 step instead of with the product's own bf16 step.

Use:  with fp8_frozen_linears(): Q.forward(P, ...)   — patches oracle.qwen25vl._lin for the decoder's q/k/v/o/gate/up/down projections."""
import contextlib

import torch
import torch.nn.functional as F

from . import qwen25vl as Q
from .kernels_ref import gemm_fp8_ref, quant_fp8_rows_ref

FROZEN = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")
# the frozen vision tower's contractions (rga3.model.qwen_train.vision_block_forward_fp8); the build pads the MLP width 3420 -> 3456 with zero columns, which changes
# neither a row's amax nor any code, so the down projection is quantised here at its own width
FROZEN_VISION = ("attn.qkv", "attn.proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


class Fp8FrozenLinear(torch.autograd.Function):
    """y = bf16( e4m3(x) . e4m3(W)^T * sx * sw + bias );  dx = bf16( e4m3(dy) . e4m3(W^T)^T * sdy * swt ).  W, bias frozen."""

    @staticmethod
    def forward(ctx, x, w, bias):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).to(torch.bfloat16).float()      # the build quantises its bf16 activation
        qx, sx = quant_fp8_rows_ref(x2)
        qw, sw = quant_fp8_rows_ref(w)
        ctx.save_for_backward(w)
        ctx.shp = shp
        return gemm_fp8_ref(qx, sx, qw, sw, bias).reshape(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        d2 = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16).float()
        qd, sd = quant_fp8_rows_ref(d2)
        qwt, swt = quant_fp8_rows_ref(w.t().contiguous())          # rows of W^T = input columns of W: the build's transposed pack
        return gemm_fp8_ref(qd, sd, qwt, swt).reshape(ctx.shp), None, None


@contextlib.contextmanager
def fp8_frozen_linears(vision: bool = False):
    orig = Q._lin

    def lin(x, P, name):
        dec = name.startswith("model.layers.") and name.endswith(FROZEN) and x.shape[-1] % 128 == 0
        vis = vision and name.startswith("visual.blocks.") and name.endswith(FROZEN_VISION) and ((x.shape[-1] + 63) // 64 * 64) % 128 == 0
        if not (dec or vis):
            return orig(x, P, name)
        y = Fp8FrozenLinear.apply(x, P[name + ".weight"], P.get(name + ".bias"))
        a = P.get(name + ".lora_A.default.weight")
        if a is not None:   # the LoRA branch stays outside the e4m3 contraction (bf16 in the build, fp32 here)
            y = y + P["lora_scaling"] * F.linear(F.linear(x, a), P[name + ".lora_B.default.weight"])
        return y

    Q._lin = lin
    try:
        yield
    finally:
        Q._lin = orig


# --------------------------------------------------------------------------------------------------------------------------------------------------------------
# The same step on the PRODUCT'S OWN e4m3 operand codes.  An activation that differs by one bf16 ulp between the two sides lands on the
# neighbouring e4m3 code (12.5 % apart) in ~3 % of its elements; through eight contractions that alone moves the layer's gradients by ~10 % and hides any kernel
# error below it.  Here every frozen contraction of the oracle multiplies the codes (and row scales) the build itself quantised -- captured through
# rga3.model.qwen_train._fp8_tap -- with weight codes the oracle quantises itself (bit-exact against rga3_quant_fp8_rows, tests/test_kernels_gpu.py), in the build's
# grouping: q | k | v and gate | up are ONE contraction each (one activation row scale forward; backward one scale per row of the concatenated gradient and per row of
# the concatenated W^T).  What is left between the two sides is the kernels' own arithmetic.
class _CodesLinear(torch.autograd.Function):
    """ys = bf16(e4m3 codes . e4m3(W_i)^T) for the weights W_i of one fused contraction; dx = bf16(e4m3 gradient codes . e4m3([W_1; W_2; ..]^T)^T).
    codes_f / codes_b: (uint8 codes viewed as float8_e4m3fn [T, K] / [T, sum N_i], f32 row scales [T]) from the build; `order_b` maps the build's gradient
    column order onto the concatenation [W_1; W_2; ...] (gate | up travel in interleaved 16-column blocks)."""

    @staticmethod
    def forward(ctx, x, codes_f, codes_b, order_b, n_w, *wb):
        ws, bs = wb[:n_w], wb[n_w:]
        qx, sx = codes_f
        ctx.save_for_backward(*ws)
        ctx.codes_b, ctx.order_b, ctx.shp = codes_b, order_b, x.shape
        outs = []
        for w, b in zip(ws, bs):
            qw, sw = quant_fp8_rows_ref(w)
            outs.append(gemm_fp8_ref(qx, sx, qw, sw, b).reshape(*x.shape[:-1], w.shape[0]))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        ws = ctx.saved_tensors
        qd, sd = ctx.codes_b
        wt = torch.cat(list(ws), 0).t().contiguous()              # [K, sum N_i]: rows of the concatenated W^T, quantised jointly as the build's transposed pack
        if ctx.order_b is not None:
            wt = wt[:, ctx.order_b]
        qwt, swt = quant_fp8_rows_ref(wt)
        dx = gemm_fp8_ref(qd, sd, qwt, swt).reshape(ctx.shp)
        return (dx, None, None, None, None) + (None,) * (2 * len(ws))


def _as_codes(rec):
    q, s = rec
    return q.detach().cpu().contiguous().view(torch.float8_e4m3fn), s.detach().float().cpu()


@contextlib.contextmanager
def fp8_frozen_linears_on_codes(codes, inter_pad=None):
    """codes: {key: (uint8 codes, scales)} as rga3.model.qwen_train._fp8_tap saw them during ONE decoder layer's forward + backward (keys wqkv, wo, wgu, wd and
    their transposed-pack twins wqkv_t, wo_t, wgu_t, wd_t).  Patches oracle.qwen25vl._lin like fp8_frozen_linears."""
    orig = Q._lin
    stash = {}

    def fused(x, P, names, kf, kb, order_b=None):
        ws = [P[n + ".weight"] for n in names]
        bs = [P.get(n + ".bias") for n in names]
        outs = _CodesLinear.apply(x, _as_codes(codes[kf]), _as_codes(codes[kb]), order_b, len(ws), *ws, *bs)
        res = {}
        for n, y in zip(names, outs):
            a = P.get(n + ".lora_A.default.weight")
            if a is not None:   # the LoRA branch stays outside the e4m3 contraction (bf16 in the build, fp32 here)
                y = y + P["lora_scaling"] * F.linear(F.linear(x, a), P[n + ".lora_B.default.weight"])
            res[n] = y
        return res

    def lin(x, P, name):
        if not (name.startswith("model.layers.") and name.endswith(FROZEN)):
            return orig(x, P, name)
        pre = name[:name.index("self_attn.") if "self_attn." in name else name.index("mlp.")]
        if name.endswith("self_attn.q_proj"):
            stash.update(fused(x, P, [pre + "self_attn.q_proj", pre + "self_attn.k_proj", pre + "self_attn.v_proj"], "wqkv", "wqkv_t"))
        elif name.endswith("mlp.gate_proj"):
            I = P[name + ".weight"].shape[0]
            assert I % 16 == 0
            # the build's gradient columns: 16-column blocks alternating gate / up -> column j of the build = row order_b[j] of [Wg; Wu]
            blk = torch.arange(I).view(I // 16, 16)
            order_b = torch.stack([blk, blk + I], 1).reshape(-1)
            stash.update(fused(x, P, [pre + "mlp.gate_proj", pre + "mlp.up_proj"], "wgu", "wgu_t", order_b))
        elif name.endswith("self_attn.o_proj"):
            stash.update(fused(x, P, [name], "wo", "wo_t"))
        elif name.endswith("mlp.down_proj"):
            stash.update(fused(x, P, [name], "wd", "wd_t"))
        return stash.pop(name)

    Q._lin = lin
    try:
        yield
    finally:
        Q._lin = orig
