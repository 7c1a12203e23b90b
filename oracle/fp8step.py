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
