"""Data-parallel gradient exchange for the RGA3 training step: bucketed all-reduce over RCCL (backend "nccl" on ROCm)
launched from autograd hooks so it overlaps the rest of backward — the MI355X-native equivalent of the reference's
DeepSpeed ZeRO-2 reduce-scatter (train_joint.py:325-346).

One process per GPU (torchrun).  The frozen 7B weights are replicated (16.6 GB of 288 GB), so plain DDP is used instead of
ZeRO partitioning; only the trainable set (~1.15 B params: embed_tokens, lm_head, LoRA, mask decoder, text_hidden_fcs) is
exchanged.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are large (default 256 MB) so each collective
amortises its launch and RCCL can stripe rings across links; under gradient accumulation the exchange happens once per
optimizer step (no_sync), which ZeRO-2 cannot do.
"""
from __future__ import annotations

import contextlib
from typing import Iterable, List

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 256.0, process_group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.sync = True
        # buckets in REVERSE parameter order (gradients arrive roughly back to front)
        cap = int(bucket_mb * (1 << 20))
        self.buckets = []
        cur, size = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > cap or cur[0].dtype != p.dtype):
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self.flat, self.slices, self.pending, self.handles = [], {}, [], []
        for bi, b in enumerate(self.buckets):
            n = sum(p.numel() for p in b)
            flat = torch.zeros(n, dtype=b[0].dtype, device=b[0].device)
            self.flat.append(flat)
            off = 0
            for p in b:
                self.slices[p] = (bi, off, p.numel())
                off += p.numel()
            self.pending.append(len(b))
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._avg = self.world > 1 and dist.get_backend(process_group) == "nccl"

    def grad_view(self, p):
        bi, off, n = self.slices[p]
        return self.flat[bi][off:off + n].view_as(p)

    def flat_grads(self):
        """The flat buckets themselves: together they hold exactly the gradients of ``params`` (every element belongs to one slice),
        so a reduction over all gradients (the clipping norm) can run once per bucket instead of once per tensor."""
        return list(self.flat)

    def _on_grad(self, p):
        bi, off, n = self.slices[p]
        if p.grad is not None:   # (a graphed backward hands an undefined gradient to parameters its forward never used: nothing to add)
            if p in self._written:   # a later micro-step of the same optimizer step accumulates
                self.flat[bi][off:off + n].add_(p.grad.reshape(-1))
            else:
                self.flat[bi][off:off + n].copy_(p.grad.reshape(-1))
                self._written.add(p)
            p.grad = None  # the bucket slice is the gradient from here on (no second copy kept)
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        self._accum_started[bi] = True
        if self.sync and self.world > 1 and not self._launched[bi]:
            self._launched[bi] = True
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self.handles.append((bi, dist.all_reduce(self.flat[bi], op=op, group=self.pg, async_op=True)))

    def begin_step(self):
        """Call before the first micro-step of an optimizer step."""
        self._accum_started = [False] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._written = set()
        self.handles = []
        self.pending = [len(b) for b in self.buckets]

    def begin_micro_step(self):
        self.pending = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)

    @contextlib.contextmanager
    def no_sync(self):
        """Accumulate locally (gradient accumulation): no collective is launched for micro-steps run under this context."""
        old, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = old

    def finish(self):
        """Wait for the outstanding collectives; afterwards grad_view(p) holds the averaged gradient.  Buckets some of whose parameters got
        no gradient in this step (parameters the loss does not reach: e.g. the IoU / object-score heads of the mask decoder, which only feed an
        argmax) never count down to zero during backward: they are exchanged here (their untouched slices hold zeros)."""
        for p in self.params:    # a parameter without a gradient in this step must not hand last step's slice to the optimizer
            if p not in self._written:
                bi, off, n = self.slices[p]
                self.flat[bi][off:off + n].zero_()
        if self.sync and self.world > 1:
            for bi in range(len(self.buckets)):
                if not self._launched[bi]:
                    self._launch(bi)
        for bi, h in self.handles:
            h.wait()
            if not self._avg and self.world > 1:
                self.flat[bi].div_(self.world)
        self.handles = []

    def remove(self):
        for h in self._hooks:
            h.remove()


class FusedAdamW:
    """AdamW over (bf16 param, bf16 grad) pairs with fp32 master weights and moments in one HIP kernel per tensor, global-norm
    clipping folded into the kernel's gradient scale (train_joint.py:300-324: lr 4e-5, betas (0.9, 0.95), wd 0, clip 1.0)."""

    def __init__(self, params, lr=4e-5, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.master = [p.detach().float().clone() for p in self.params]
        self.m = [torch.zeros_like(x) for x in self.master]
        self.v = [torch.zeros_like(x) for x in self.master]
        self.t = 0

    def step(self, grad_of, flat_grads=None):
        """grad_of(p) -> gradient tensor (e.g. GradBucketReducer.grad_view).  flat_grads: optional list of flat tensors that together
        hold exactly these gradients (GradBucketReducer.flat_grads()): the clipping norm then takes one launch per bucket."""
        from ..hip import ops

        self.t += 1
        grads = [grad_of(p).contiguous() for p in self.params]
        scale = 1.0
        if self.max_norm is not None:
            acc = torch.zeros(1, dtype=torch.float32, device=self.params[0].device)
            for g in (flat_grads if flat_grads is not None else grads):
                ops.sumsq_accum_(g.reshape(-1), acc)
            norm = float(acc.sqrt())
            scale = min(1.0, self.max_norm / (norm + 1e-6))
        for p, w, g, m, v in zip(self.params, self.master, grads, self.m, self.v):
            ops.adamw_step_(p.data, w, g, m, v, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, scale)
        return scale
