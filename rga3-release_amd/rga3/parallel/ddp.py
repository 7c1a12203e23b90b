"""Data-parallel gradient exchange for the RGA3 training step: bucketed all-reduce over RCCL (backend "nccl" on ROCm)
launched from autograd hooks so it overlaps the rest of backward — the MI355X-native equivalent of the reference's
DeepSpeed ZeRO-2 reduce-scatter (train_joint.py:325-346).

One process per GPU (torchrun).  The frozen 7B weights are replicated (16.6 GB of 288 GB), so plain DDP is used instead of
ZeRO partitioning; only the trainable set (~1.15 B params: embed_tokens, lm_head, LoRA, mask decoder, text_hidden_fcs) is
exchanged.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are large (default 256 MB) so each collective
amortises its launch and RCCL can stripe rings across links; under gradient accumulation the exchange happens once per
optimizer step (no_sync), which ZeRO-2 cannot do.

Collective order is rank-independent BY CONSTRUCTION: bucket i is launched only after buckets 0..i-1, and every bucket is
launched on every rank in every synchronising step (a bucket some of whose parameters got no gradient on this rank — a
sample without [SEG] skips SAM2, a step without labelled rows drops the lm_head gradient — waits for finish(), where
the remaining buckets go out in index order with zeros in the untouched slices).  NCCL and gloo pair collectives by issue
order, so any data-dependent launch order would silently pair different buckets across ranks.

embed_tokens (545 M of the 1.15 B trainable elements) is exchanged SPARSELY: one sample touches <= S rows of the table, and
its gradient is the last one backward produces, so a dense 1.09 GB bucket could not overlap anything.  Each
rank contributes (unique row ids, summed rows) — <= 15 MB — through one all-gather, and every rank applies the N
contributions in rank order (bitwise-identical result on all ranks).
"""
from __future__ import annotations

import contextlib
import math
import os
import weakref
from typing import Iterable, List

import numpy as np
import torch
import torch.distributed as dist

from ..utils.staging import upload as _upload

_ALIGN = 8   # bucket slices start at multiples of 8 elements (16 B for bf16): the optimizer kernels use 16-byte accesses

# id(parameter) -> reducer that takes that parameter's gradient as (row ids, rows) instead of a dense tensor (see rga3.model.qwen_train.EmbedFn)
_sparse_sinks: "weakref.WeakValueDictionary[int, GradBucketReducer]" = weakref.WeakValueDictionary()


@contextlib.contextmanager
def _host_staged_collective():
    """gloo has no all-gather on device tensors: the functional multi-rank runs on ONE shared GPU (bench.py RGA3_BENCH_SHARE_GPU, tests/test_ddp_shared_gpu.py)
    stage the rows through the host, which synchronises by construction.  RCCL never takes this branch.  The staging is exempted from torch's sync debug mode so
    that a test can run the whole multi-rank step under set_sync_debug_mode("error") and still catch every OTHER device -> host wait."""
    mode = torch.cuda.get_sync_debug_mode() if torch.cuda.is_available() else 0
    if mode:
        torch.cuda.set_sync_debug_mode(0)
    try:
        yield
    finally:
        if mode:
            torch.cuda.set_sync_debug_mode(mode)


def sparse_sink_for(param):
    return _sparse_sinks.get(id(param))


_dense_sinks: "weakref.WeakValueDictionary[int, GradBucketReducer]" = weakref.WeakValueDictionary()
_DIRECT_GRAD = os.environ.get("RGA3_DIRECT_GRAD", "1") != "0"     # A/B switch


def dense_grad_out_for(param):
    """The bucket slice a producer may write ``param``'s gradient INTO (shaped like the parameter), or None.  The LM head's dW is 1.09 GB: produced as a fresh tensor
    it was copied into its bucket afterwards (2.2 GB of traffic per step); written in place the copy disappears.  Only for the FIRST gradient of an optimizer step
    (later micro-steps must add), and only when dtype and device match; the producer then returns that very tensor as the gradient, and _flush recognises it."""
    red = _dense_sinks.get(id(param))
    if red is None or red._removed or not _DIRECT_GRAD or param in red._written or param not in red._view:
        return None
    v = red._view[param]
    if v.dtype != param.dtype or v.device != param.device:
        return None
    return v.view(param.shape)


def sparse_candidates(model):
    """Parameters whose gradient may be exchanged as rows: the input embedding table, unless it is tied to the LM head (Qwen2.5-VL-3B,
    tie_word_embeddings): then the head's dense dW lands in the same tensor and the table goes through a dense bucket like everything else."""
    emb = model.get_input_embeddings().weight
    head = model.get_output_embeddings().weight
    if not emb.requires_grad or emb is head:
        return []
    return [emb]


class GradBucketReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 256.0, process_group=None, sparse_params=(), sparse: bool = True,
                 announce_cap: int = 8192):
        self.pg = process_group
        self._removed = False
        # rows per rank that the forward-time announcement carries (ids, not only their number): with them every rank knows ALL ranks' row ids on the host
        # before backward ends, so finish() needs neither a second id all-gather nor a device-side unique (a data-dependent shape = a device -> host wait
        # with the optimizer's launches still to be issued).  A step whose union outgrows it falls back to the exchange-at-finish path.  Multiple of 8.
        self._ann_cap = max(8, (int(announce_cap) + 7) // 8 * 8)
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.sync = True
        self.sparse_params: List[torch.nn.Parameter] = [p for p in sparse_params if p.requires_grad] if sparse else []
        sp_ids = {id(p) for p in self.sparse_params}
        self.all_params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.params: List[torch.nn.Parameter] = [p for p in self.all_params if id(p) not in sp_ids]
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
        self._stash = [[] for _ in self.buckets]
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self.slices[p] = (bi, off, p.numel())
                off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            self.flat.append(torch.zeros(off, dtype=b[0].dtype, device=b[0].device))
            self.pending.append(len(b))
        self._view = {p: self.flat[self.slices[p][0]][self.slices[p][1]:self.slices[p][1] + self.slices[p][2]] for p in self.params}   # built once: the hook runs ~300 times per step
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        for p in self.params:
            _dense_sinks[id(p)] = self
        self._nccl = self.world > 1 and dist.get_backend(process_group) == "nccl"
        self._avg = self._nccl
        # ---- sparse parameters: a persistent dense gradient buffer (zero outside the rows touched this step) + the exchange state
        self._sp = {}
        for p in self.sparse_params:
            assert p.dim() == 2
            self._sp[id(p)] = {"p": p, "dense": torch.zeros_like(p.data), "union": np.zeros(0, dtype=np.int64), "last_ids": None, "counts": None}
            _sparse_sinks[id(p)] = self
            # a sparse parameter must receive its gradient ONLY as rows: a dense gradient (tied word embeddings: lm_head.weight IS embed_tokens.weight, so the
            # LM head's dW accumulates into the same .grad) would never be exchanged, never reach the optimizer and never be cleared.  Refuse it loudly; the
            # caller registers such a parameter as dense (rga3.parallel.ddp.sparse_candidates does that).
            self._hooks.append(p.register_post_accumulate_grad_hook(self._dense_on_sparse))
        self.sparse_bytes_last = 0
        self.begin_step()

    @staticmethod
    def _dense_on_sparse(p):
        if p.grad is not None:
            raise RuntimeError("GradBucketReducer: a parameter registered as sparse (row-wise gradient) received a dense gradient -- with tied word embeddings "
                               "(lm_head.weight is embed_tokens.weight) register it as a dense parameter (sparse_candidates(model))")

    # ---------------------------------------------------------------------------------------------- views for the optimizer
    def grad_view(self, p):
        st = self._sp.get(id(p))
        if st is not None:
            return st["dense"]
        bi, off, n = self.slices[p]
        return self.flat[bi][off:off + n].view_as(p)

    def flat_grads(self):
        """Flat tensors that together hold exactly the gradients of ``params`` (padding elements are zero), so a reduction over all gradients
        (the clipping norm) can run once per bucket instead of once per tensor."""
        return list(self.flat) + [st["dense"].view(-1) for st in self._sp.values()]

    def layout(self):
        """[(flat gradient tensor, [(param, offset, numel), ...]), ...]: the buckets and the dense buffers of the sparse parameters.  FusedAdamW.for_reducer
        lays its fp32 state (and the bf16 parameters themselves) out the same way, so the optimizer is one launch per bucket instead of one per tensor."""
        out = []
        for bi, b in enumerate(self.buckets):
            out.append((self.flat[bi], [(p, self.slices[p][1], p.numel()) for p in b]))
        for st in self._sp.values():
            out.append((st["dense"].view(-1), [(st["p"], 0, st["p"].numel())]))
        return out

    # ---------------------------------------------------------------------------------------------- dense buckets
    def _on_grad(self, p):
        g = p.grad
        bi = self.slices[p][0]
        if g is not None:   # (a graphed backward hands an undefined gradient to parameters its forward never used: nothing to add)
            # parked until the bucket's last gradient arrives, then copied with ONE multi-tensor launch per bucket (_flush): the ~250 per-parameter
            # copies of a step were ~250 launches of a few microseconds each in the launch-bound mask-path backward (rocprofv3, tools/step_timeline.py)
            self._stash[bi].append((p, g.reshape(-1)))
            p.grad = None  # the bucket slice is the gradient from here on (no second copy kept)
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._flush(bi)
            self._launch_ready()

    def _flush(self, bi):
        """Move the parked gradients of bucket bi into its flat buffer: first arrival of an optimizer step copies, later micro-steps add."""
        st = self._stash[bi]
        if not st:
            return
        cp_v, cp_g, ad_v, ad_g = [], [], [], []
        for p, g in st:
            view = self._view[p]
            if p in self._written:
                ad_v.append(view)
                ad_g.append(g if g.dtype == view.dtype else g.to(view.dtype))
            else:
                self._written.add(p)
                if g.data_ptr() == view.data_ptr() and g.dtype == view.dtype:
                    pass        # produced in place (dense_grad_out_for): nothing to move
                elif g.is_cuda and g.dtype == view.dtype:
                    cp_v.append(view)
                    cp_g.append(g)
                else:
                    view.copy_(g)
        # multi-tensor kernels, not copy_() per tensor: a device-to-device copy_ goes through hipMemcpyAsync, whose blit dispatch left the GPU idle ~37 us each
        if cp_v:
            torch._foreach_copy_(cp_v, cp_g)
        if ad_v:
            torch._foreach_add_(ad_v, ad_g)
        st.clear()

    def _launch_ready(self):
        """Launch, in index order, every bucket whose gradients are all in: bucket i never goes out before bucket i-1 (rank-independent order)."""
        while self._next < len(self.buckets) and self.pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi):
        if self.sync and self.world > 1:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            if self._nccl or not self.flat[bi].is_cuda:
                self.handles.append((bi, dist.all_reduce(self.flat[bi], op=op, group=self.pg, async_op=True)))
            else:     # gloo on device tensors (shared-GPU functional runs): its transfers are staged through the host and synchronise by construction --
                      # in a worker thread, so the collective is run to completion inside the exemption (functional path: no overlap to lose)
                with _host_staged_collective():
                    dist.all_reduce(self.flat[bi], op=op, group=self.pg)
                self.handles.append((bi, None))

    def begin_step(self):
        """Call before the first micro-step of an optimizer step."""
        for st in self._stash:   # gradients of a step that was never finished are dropped with it
            st.clear()
        self._written = set()
        self.handles = []
        self.begin_micro_step()
        for st in self._sp.values():
            self._clear_rows(st)

    def begin_micro_step(self):
        for bi in range(len(self.buckets)):   # a bucket the previous micro-step left incomplete: its gradients go in before the next ones are parked
            self._flush(bi)
        self.pending = [len(b) for b in self.buckets]
        self._next = 0

    @contextlib.contextmanager
    def no_sync(self):
        """Accumulate locally (gradient accumulation): no collective is launched for micro-steps run under this context."""
        old, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = old

    def finish(self):
        """Launch what backward left (buckets holding a parameter without a gradient on this rank: their untouched slices are zeroed first), exchange
        the sparse rows, wait for everything; afterwards grad_view(p) holds the averaged gradient."""
        for bi in range(len(self.buckets)):
            self._flush(bi)
        if len(self._written) != len(self.params):
            for p in self.params:    # a parameter without a gradient in this step must not hand last step's slice to the optimizer
                if p not in self._written:
                    self._view[p].zero_()
        if self.sync:
            while self._next < len(self.buckets):
                self._launch(self._next)
                self._next += 1
            for st in self._sp.values():
                self._exchange_sparse(st)
        else:   # a local-only step (finish() under no_sync): nothing is exchanged; the rows touched are still remembered for the clean-up
            for st in self._sp.values():
                mine = st.get("dev_ids") or []
                st["last_ids"] = mine[0] if len(mine) == 1 and mine[0].numel() == st["union"].size else _upload(st["union"], st["dense"].device)
        for bi, h in self.handles:
            if h is not None:
                h.wait()          # RCCL: a stream dependency, no host wait
            if not self._avg and self.world > 1:
                self.flat[bi].div_(self.world)
        self.handles = []

    # ---------------------------------------------------------------------------------------------- sparse rows
    def _clear_rows(self, st):
        """Zero the rows of the dense buffer that the previous optimizer step touched (the buffer is zero everywhere else)."""
        if st["last_ids"] is not None and st["last_ids"].numel():
            from ..hip import ops
            d = st["dense"]
            if d.is_cuda:
                ops.scatter_rows_(d, st["last_ids"], torch.zeros((st["last_ids"].numel(), d.shape[1]), dtype=d.dtype, device=d.device))
            else:
                d[st["last_ids"]] = 0
        st["last_ids"] = None
        st["union"] = np.zeros(0, dtype=np.int64)
        st["dev_ids"] = []

    def announce_sparse(self, p, uniq_ids_np):
        """Forward of a micro-step: the rows this rank will contribute are known on the host.  The running union (its ids and their number) is exchanged
        now, on a side stream, so that finish() can size the row all-gather, address every rank's rows and name the union of all ranks' rows without
        waiting for the GPU.  Message per rank: [ids (announce_cap, padded with the scratch row id V) | count | 7 x pad] int64."""
        st = self._sp[id(p)]
        st["union"] = np.union1d(st["union"], uniq_ids_np)
        if self.world == 1:
            return
        dev = p.device
        cap, n_ids = self._ann_cap, int(st["union"].size)
        msg = torch.full((cap + 8,), int(p.shape[0]), dtype=torch.int64)
        msg[:min(n_ids, cap)] = torch.from_numpy(st["union"][:cap])
        msg[cap] = n_ids
        if dev.type == "cuda" and self._nccl:
            side = st.setdefault("side", torch.cuda.Stream(device=dev))
            msg = msg.pin_memory()
            host = torch.empty((self.world, cap + 8), dtype=torch.int64).pin_memory()
            with torch.cuda.stream(side):
                nd = msg.to(dev, non_blocking=True)
                out = torch.empty((self.world, cap + 8), dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(out.view(-1), nd, group=self.pg)
                host.copy_(out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            st["counts"] = (host, ev, (msg, nd, out))
        else:
            host = torch.empty((self.world, cap + 8), dtype=torch.int64)
            dist.all_gather_into_tensor(host.view(-1), msg, group=self.pg)
            st["counts"] = (host, None, None)

    def add_sparse(self, p, ids_dev, rows):
        """Backward of a micro-step: accumulate (unique row ids, summed rows) of this rank into the dense buffer."""
        st = self._sp[id(p)]
        _scatter_add(st["dense"], ids_dev, rows, 1.0)
        st.setdefault("dev_ids", []).append(ids_dev)

    def _exchange_sparse(self, st):
        d = st["dense"]
        dev = d.device
        ids_np = st["union"]
        mine = st.get("dev_ids") or []
        # the union of this step's row ids on the device: with one micro-step it is the tensor backward handed over (an upload of the host copy would be a
        # stream sync at the end of backward, with the optimizer's launches still to be issued)
        ids = mine[0] if len(mine) == 1 and mine[0].numel() == ids_np.size else _upload(ids_np, dev)
        if self.world == 1:
            st["last_ids"] = ids
            self.sparse_bytes_last = 0
            return
        host, ev, keep = st["counts"]
        if ev is not None:
            ev.synchronize()      # recorded during forward: long done when backward ends
        acap = self._ann_cap
        counts = [int(v) for v in host[:, acap].tolist()]
        H = d.shape[1]
        cap = (max(counts) + 63) // 64 * 64
        announced = max(counts) <= acap      # every rank's ids travelled with the announcement (the same decision on every rank: same gathered counts)
        if announced:
            if keep is not None:             # RCCL: the gathered ids are on the device already (side stream) -- this stream only has to wait for them
                torch.cuda.current_stream(dev).wait_event(ev)
                ids_all = keep[2]
                ids_all.record_stream(torch.cuda.current_stream(dev))
            else:
                ids_all = host if dev.type != "cuda" else _upload(host, dev)
            rank_ids = [ids_all[r, :counts[r]] for r in range(self.world)]
            ids = rank_ids[dist.get_rank(self.pg)]     # this rank's union, as announced (sorted unique ids)
        send_rows = torch.zeros((cap, H), dtype=d.dtype, device=dev)
        if ids.numel():
            send_rows[:ids.numel()] = _gather(d, ids)
        all_rows = torch.empty((self.world * cap, H), dtype=d.dtype, device=dev)
        id_bytes = self.world * (acap + 8) * 8
        if not announced:
            # fallback (a union larger than announce_cap): ids go out here, padded with the scratch row id V, and the union is taken on the device
            send_ids = torch.full((cap,), d.shape[0], dtype=torch.int64, device=dev)
            send_ids[:ids.numel()] = ids
            all_ids = torch.empty((self.world * cap,), dtype=torch.int64, device=dev)
            id_bytes += self.world * cap * 8
        if self._nccl or dev.type != "cuda":
            if not announced:
                dist.all_gather_into_tensor(all_ids, send_ids, group=self.pg)
            dist.all_gather_into_tensor(all_rows, send_rows, group=self.pg)
        else:   # gloo has no all-gather on device tensors (functional multi-rank runs on a shared GPU, bench.py RGA3_BENCH_SHARE_GPU): stage through the host
            with _host_staged_collective():
                if not announced:
                    hi = torch.empty(all_ids.shape, dtype=all_ids.dtype)
                    dist.all_gather_into_tensor(hi, send_ids.cpu(), group=self.pg)
                    all_ids.copy_(hi)
                hr = torch.empty(all_rows.shape, dtype=all_rows.dtype)
                dist.all_gather_into_tensor(hr, send_rows.cpu(), group=self.pg)
                all_rows.copy_(hr)
        if not announced:
            rank_ids = [all_ids[r * cap:r * cap + counts[r]] for r in range(self.world)]
        self.sparse_bytes_last = id_bytes + self.world * cap * H * d.element_size()
        # local rows out, then the N contributions in rank order (every id list is unique within itself): identical bits on every rank
        if ids.numel():
            _zero_rows(d, ids)
        touched = []
        for r in range(self.world):
            if counts[r]:
                _scatter_add(d, rank_ids[r], all_rows[r * cap:r * cap + counts[r]], 1.0 / self.world)
                touched.append(rank_ids[r])
        # the union of all ranks' rows (unique, sorted: the optimizer sums the gradient norm over exactly these rows): from the announced host copy -- no
        # device-side unique, whose data-dependent shape would stall the host at the end of backward
        if announced:
            hn = host.numpy()
            parts = [hn[r, :counts[r]] for r in range(self.world) if counts[r]]
            st["last_ids"] = _upload(np.unique(np.concatenate(parts)), dev) if parts else None
        else:
            st["last_ids"] = torch.unique(torch.cat(touched)) if touched else None

    def remove(self):
        """Detach from the parameters: hooks off, and neither sink table points here any more (a producer that asked dense_grad_out_for() after this would
        otherwise keep writing dW into a bucket nobody tracks: with the hooks gone ``_written`` never changes, so every backward would return the SAME memory
        as a fresh gradient and autograd's accumulation would add a tensor to itself)."""
        self._removed = True
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.sparse_params:
            if _sparse_sinks.get(id(p)) is self:
                del _sparse_sinks[id(p)]
        for p in self.params:
            if _dense_sinks.get(id(p)) is self:
                del _dense_sinks[id(p)]


def _gather(d, ids):
    if d.is_cuda:
        from ..hip import ops
        return ops.gather_rows(d, ids)
    return d[ids]


def _zero_rows(d, ids):
    if d.is_cuda:
        from ..hip import ops
        ops.scatter_rows_(d, ids, torch.zeros((ids.numel(), d.shape[1]), dtype=d.dtype, device=d.device))
    else:
        d[ids] = 0


def _scatter_add(d, ids, rows, scale):
    """d[ids[i]] += scale * rows[i]; ids unique within the call."""
    if ids.numel() == 0:
        return
    if d.is_cuda:
        from ..hip import ops
        ops.scatter_add_rows_(d, ids, rows, scale)
    else:   # CPU tests (gloo): same arithmetic order — one rounding to the storage dtype per contribution
        d[ids] = (d[ids].float() + rows.float() * scale).to(d.dtype)


class WarmupCosineLR:
    """DeepSpeed's WarmupCosineLR as configured by reference train_joint.py:308-317 (total_num_steps = epochs * steps_per_epoch, warmup_min_ratio 0,
    cos_min_ratio 0.03, warmup_num_steps = 3 % of the total, warmup_type "linear").  DeepSpeed 0.16.3 is not in this image: restated from its
    published lr_schedules.py (parity unpinned).  The engine steps the scheduler AFTER each optimizer step and the scheduler does not step at
    construction, so optimizer step 0 runs at the configured lr and step k >= 1 at lr * ratio(last_batch_iteration = k - 1)."""

    def __init__(self, total_num_steps: int, warmup_num_steps: int, warmup_min_ratio: float = 0.0, cos_min_ratio: float = 0.03, warmup_type: str = "linear"):
        self.total, self.warm = int(total_num_steps), max(2, int(warmup_num_steps))
        self.warm_min, self.cos_min, self.warmup_type = float(warmup_min_ratio), float(cos_min_ratio), warmup_type
        if warmup_type not in ("linear", "log"):
            raise ValueError(f"warmup_type {warmup_type!r}")

    def ratio(self, it: int) -> float:
        if it < 0:
            return 0.0
        if it < self.warm:
            r = it / self.warm if self.warmup_type == "linear" else math.log(it + 1) / math.log(self.warm)
            return self.warm_min + r * (1.0 - self.warm_min)
        real_last, real_total = it - self.warm + 1, self.total - self.warm
        r = (1.0 + math.cos(math.pi * real_last / real_total)) / 2.0
        return max(0.0, self.cos_min + (1.0 - self.cos_min) * r)

    def scale_at(self, step: int) -> float:
        """Factor on the configured lr for optimizer step ``step`` (0-based)."""
        return 1.0 if step == 0 else self.ratio(step - 1)


class FusedAdamW:
    """AdamW over (bf16 param, bf16 grad) pairs with fp32 master weights and moments in one HIP kernel per tensor, global-norm
    clipping folded into the kernel's gradient scale (train_joint.py:300-324: lr 4e-5, betas (0.9, 0.95), wd 0, clip 1.0).  The clipping norm
    never visits the host: a deterministic two-stage sum of squares leaves it in device memory and the update kernel derives its scale from it."""

    def __init__(self, params, lr=4e-5, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, schedule: WarmupCosineLR = None, layout=None):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.schedule = schedule
        self._flat = None
        self._sparse_src, self._row_mask = {}, {}
        self.health_every, self._health = 16, None     # poll the GEMM give-up counters every N steps (0 = off)
        if layout is not None:
            self._init_flat(layout)
        else:
            self.master = [p.detach().float().clone() for p in self.params]
            self.m = [torch.zeros_like(x) for x in self.master]
            self.v = [torch.zeros_like(x) for x in self.master]
        self.t = 0
        dev = self.params[0].device
        self._acc = torch.zeros(1, dtype=torch.float32, device=dev)
        self._partials = torch.zeros(2048, dtype=torch.float32, device=dev)

    @classmethod
    def for_reducer(cls, reducer: "GradBucketReducer", **kw):
        """Optimizer state laid out like the reducer's gradient buckets: ONE AdamW launch per bucket.  (Per-tensor launches left the GPU idle between the
        ~130 small kernels of a step: the host needs ~20 us per launch, a LoRA factor's update takes 3 us -- 2.9 ms of a 170 ms step.)  The bf16 parameters of a
        multi-tensor bucket are re-pointed at slices of one flat buffer (values unchanged), as DDP / FSDP flat parameters are."""
        opt = cls(reducer.all_params, layout=reducer.layout(), **kw)
        # tables whose gradient arrives as (row ids, rows): AdamW then touches only the rows that ever received a gradient (exact without weight decay, see step())
        opt._sparse_src = {id(st["p"]): st for st in reducer._sp.values()}
        return opt

    def _init_flat(self, layout):
        order = {id(p): i for i, p in enumerate(self.params)}
        self.master, self.m, self.v = [None] * len(self.params), [None] * len(self.params), [None] * len(self.params)
        self._flat = []
        self._flat_single = []     # the one 2-D parameter a flat entry consists of (None for multi-tensor buckets)
        with torch.no_grad():
            for flat_g, items in layout:
                n = flat_g.numel()
                dev = flat_g.device
                if len(items) == 1 and items[0][1] == 0 and items[0][2] == n and items[0][0].is_contiguous():
                    flat_p = items[0][0].data.view(-1)                      # a bucket that IS one tensor (embed_tokens, lm_head): no copy
                else:
                    flat_p = torch.zeros(n, dtype=flat_g.dtype, device=dev)
                    for p, off, k in items:
                        flat_p[off:off + k].copy_(p.data.reshape(-1))
                        p.data = flat_p[off:off + k].view(p.shape)
                fm = flat_p.float()
                fmm, fv = torch.zeros_like(fm), torch.zeros_like(fm)
                self._flat.append((flat_p, fm, flat_g, fmm, fv))
                self._flat_single.append(items[0][0] if len(items) == 1 and items[0][0].dim() == 2 and items[0][2] == n else None)
                for p, off, k in items:
                    i = order[id(p)]
                    self.master[i], self.m[i], self.v[i] = fm[off:off + k].view(p.shape), fmm[off:off + k].view(p.shape), fv[off:off + k].view(p.shape)
        assert all(x is not None for x in self.master), "layout does not cover every trainable parameter"

    def current_lr(self) -> float:
        return self.lr * self.schedule.scale_at(self.t) if self.schedule is not None else self.lr

    def resync_master(self):
        """Re-read the fp32 master weights from the (bf16) parameters — call after loading a checkpoint into the model once the optimizer exists."""
        with torch.no_grad():
            for w, p in zip(self.master, self.params):
                w.copy_(p.detach().float())

    def state_dict(self):
        """Optimizer state for resume (reference: DeepSpeed save_checkpoint / load_checkpoint with optimizer + scheduler state, train_joint.py:352-366, 426-461)."""
        return {"t": self.t, "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "max_grad_norm": self.max_norm,
                "schedule": None if self.schedule is None else dict(total=self.schedule.total, warm=self.schedule.warm, warm_min=self.schedule.warm_min,
                                                                    cos_min=self.schedule.cos_min, warmup_type=self.schedule.warmup_type),
                "master": [w.detach().cpu() for w in self.master], "m": [x.detach().cpu() for x in self.m], "v": [x.detach().cpu() for x in self.v]}

    def load_state_dict(self, sd, write_params: bool = True):
        self._row_mask = {}     # rebuilt from the loaded moments at the next step
        if len(sd["master"]) != len(self.params) or any(a.shape != b.shape for a, b in zip(sd["master"], self.master)):
            raise ValueError("optimizer state does not match the parameter list (count or shapes differ)")
        self.t, self.lr, self.betas, self.eps, self.wd, self.max_norm = int(sd["t"]), sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"], sd["max_grad_norm"]
        sc = sd.get("schedule")
        self.schedule = None if sc is None else WarmupCosineLR(sc["total"], sc["warm"], sc["warm_min"], sc["cos_min"], sc["warmup_type"])
        with torch.no_grad():
            for dst, src in ((self.master, sd["master"]), (self.m, sd["m"]), (self.v, sd["v"])):
                for a, b in zip(dst, src):
                    a.copy_(b)
            if write_params:   # the bf16 parameters are the rounded masters
                for p, w in zip(self.params, self.master):
                    p.data.copy_(w.to(p.dtype))

    def grad_norm(self) -> torch.Tensor:
        """Global gradient norm of the last step() (device scalar, no sync)."""
        return self._acc.sqrt()

    def step(self, grad_of, flat_grads=None):
        """grad_of(p) -> gradient tensor (e.g. GradBucketReducer.grad_view).  flat_grads: optional list of flat tensors that together
        hold exactly these gradients (GradBucketReducer.flat_grads()): the clipping norm then takes one launch per bucket."""
        from ..hip import ops

        lr = self.current_lr()
        self.t += 1
        clip = self.max_norm is not None
        if self.params[0].is_cuda and self.health_every:     # a stream-K hand-off that gave up must not go unnoticed in a real run (no sync: GemmHealthWatch)
            if self._health is None:
                self._health = ops.GemmHealthWatch(self.health_every)
            self._health.poll()
        if self._flat is not None:   # one launch per bucket (gradients are the reducer's flat buckets themselves)
            if clip:
                for i, (_, _, g, _, _) in enumerate(self._flat):
                    p1 = self._flat_single[i]
                    st = self._sparse_src.get(id(p1)) if p1 is not None else None
                    ids = st.get("last_ids") if st is not None else None
                    if ids is not None:      # the dense buffer of a sparse table is zero outside this step's rows: sum those (<= S rows instead of 1.09 GB)
                        g = ops.gather_rows(g.view(p1.shape), ids).view(-1) if ids.numel() else g[:8]
                    ops.sumsq_det_(g, self._partials, self._acc, accumulate=i > 0)
            for i, (fp, fm, g, mm, vv) in enumerate(self._flat):
                p1 = self._flat_single[i]
                st = self._sparse_src.get(id(p1)) if p1 is not None else None
                if st is not None and self.wd == 0.0 and p1.shape[1] % 8 == 0:
                    # embed_tokens: only rows that ever received a gradient are updated.  A never-touched row has g = m = v = 0, and without weight decay
                    # AdamW leaves it bit-for-bit unchanged, so this equals the dense update (tests/test_train_gpu.py) while streaming <= S of 152 064 rows.
                    mask = self._row_mask.get(i)
                    if mask is None:
                        mask = self._row_mask[i] = self._rows_with_state(mm, vv, p1.shape)
                    ids = st.get("last_ids")
                    if ids is not None:
                        if ids.numel():
                            mask.index_fill_(0, ids, 1)
                        ops.adamw_step_clip_rows_(fp.view(p1.shape), fm.view(p1.shape), g.view(p1.shape), mm.view(p1.shape), vv.view(p1.shape), mask, lr, self.betas[0],
                                                  self.betas[1], self.eps, self.t, self._acc if clip else None, self.max_norm if clip else 0.0)
                        continue
                    self._row_mask.pop(i)     # the rows of this step are unknown (finish() was not called): dense update, the mask is rebuilt from the moments
                ops.adamw_step_clip_(fp, fm, g, mm, vv, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, self._acc if clip else None,
                                     self.max_norm if clip else 0.0)
            self._bump_versions()
            return
        grads = [grad_of(p).contiguous() for p in self.params]
        if clip:
            first = True
            for g in (flat_grads if flat_grads is not None else grads):
                ops.sumsq_det_(g.reshape(-1), self._partials, self._acc, accumulate=not first)
                first = False
        for p, w, g, m, v in zip(self.params, self.master, grads, self.m, self.v):
            ops.adamw_step_clip_(p.data, w, g, m, v, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                 self._acc if clip else None, self.max_norm if clip else 0.0)
        self._bump_versions()

    @staticmethod
    def _rows_with_state(m, v, shape):
        """uint8 [rows]: 1 where a row of the moments is not all zero (fresh optimizer: all 0; after load_state_dict: the rows trained so far)."""
        return ((m.view(shape) != 0).any(1) | (v.view(shape) != 0).any(1)).to(torch.uint8)

    def _bump_versions(self):
        """The update kernels write the parameters through raw pointers; derived tensors cached by (data_ptr, version) elsewhere (the ConvTranspose weight
        packs of the mask decoder's inference path, captured decode graphs) must see that the values changed."""
        for p in self.params:
            torch.autograd.graph.increment_version(p)
