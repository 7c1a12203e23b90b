"""ORACLE support — test infrastructure only.  Emulates bf16 STORAGE of activations and of their gradients inside the fp32 SAM2 restatement
(oracle/sam2.py): the output of every linear / LayerNorm / attention / transposed-conv is rounded to bf16 in forward, and the gradient arriving at it
is rounded to bf16 in backward.  Arithmetic inside each op stays fp32, so this is a LOWER bound on what any bf16 pipeline — this build's HIP path or the
reference under `--precision bf16` (train_joint.py:45,166-167) — loses against fp32 autograd.

Use: the gradient-parity test measures, per tensor, how far this emulation drifts from the fp32 oracle and requires the HIP path to stay within a small
multiple of that drift.  (Measured on the tiny joint fixture: the emulation alone shows 3-5 % rel-L2 on the mask decoder's transformer tensors — the
mask-loss gradient is a difference of large, nearly cancelling pixel sums — so a flat 3e-2 bound cannot hold for a bf16 backward there.)"""
import contextlib

import torch

from . import sam2 as S

_OPS = ("lin", "layer_norm", "layer_norm_2d", "sdpa", "conv_transpose_2x2")


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


@contextlib.contextmanager
def bf16_storage():
    saved = {n: getattr(S, n) for n in _OPS}

    def wrap(f):
        return lambda *a, **k: _RoundBF16.apply(f(*a, **k))

    try:
        for n, f in saved.items():
            setattr(S, n, wrap(f))
        yield
    finally:
        for n, f in saved.items():
            setattr(S, n, f)
