"""Host <-> device plumbing for the integer inputs of the hot path (token ids, masks, labels, index tables).

The reference's trainer builds every batch on the CPU (collate_fn, reference utils/dataset.py:41-134) and moves it with dict_to_cuda
(utils/utils.py:177-187) right before model(**input_dict) (train_joint.py:500-519).  The forward then needs the INTEGER inputs on the host again
(mRoPE positions, packing, cu_seqlens, CE targets, [SEG] rows are host-side integer plumbing): reading them back from the device is a device -> host
sync per step, and uploading each derived index array from pageable memory is another.  Two rules remove both:

* `dict_to_cuda` (same semantics as the reference's) keeps a PRIVATE host copy of every integer tensor (the pinned staging copy the H2D transfer
  reads, or a clone when the caller's tensor is already pinned) as a companion of the device tensor it returns (`host_of`): the plan is built from
  exactly the bytes that went to the device, no read-back, and later writes into the caller's buffer cannot reach it;
* every derived index array goes up through `upload`: one pinned staging block + a non-blocking copy on the current stream (ordered before its
  consumers like any launch; the caching host allocator holds the block until the copy has run).
"""
from __future__ import annotations

import numpy as np
import torch

_HOST_ATTR = "_rga3_host"


def attach_host(dev_t: torch.Tensor, cpu_t: torch.Tensor) -> torch.Tensor:
    """Remember the CPU tensor `dev_t` was copied from.  The companion is trusted only while both tensors are unmodified (version counters)."""
    setattr(dev_t, _HOST_ATTR, (cpu_t, cpu_t._version, dev_t._version))
    return dev_t


def host_of(t) -> np.ndarray | None:
    """numpy view of `t` on the host: the tensor itself if it lives on the CPU, its attached companion if one is valid (no sync), otherwise a
    device -> host read (a stream sync)."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t
    if not isinstance(t, torch.Tensor):
        return np.asarray(t)
    if not t.is_cuda:
        return t.detach().numpy()
    comp = getattr(t, _HOST_ATTR, None)
    if comp is not None:
        c, vc, vd = comp
        if c._version == vc and t._version == vd and tuple(c.shape) == tuple(t.shape):
            return c.detach().numpy()
    return t.detach().cpu().numpy()


def has_host(t) -> bool:
    comp = getattr(t, _HOST_ATTR, None) if isinstance(t, torch.Tensor) else None
    return comp is not None and comp[0]._version == comp[1] and t._version == comp[2]


def upload(arr, dev, dtype=None) -> torch.Tensor:
    """Host array -> device tensor without stalling the host: pinned staging + non-blocking copy on the current stream."""
    t = torch.from_numpy(np.ascontiguousarray(arr)) if isinstance(arr, np.ndarray) else arr
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    dev = torch.device(dev)
    if dev.type != "cuda":
        return t.to(dev)
    if t.numel() == 0:
        return torch.empty(t.shape, dtype=t.dtype, device=dev)
    return t.pin_memory().to(dev, non_blocking=True)


def dict_to_cuda(input_dict: dict, device=None) -> dict:
    """reference utils/utils.py:177-187 (tensors and lists of tensors -> the GPU, non-blocking), plus: the CPU tensor stays attached to its device copy
    so the forward's host-side integer plumbing reads the collate function's copy instead of the device (`host_of`).  Floating tensors are pinned first
    so `non_blocking` is real; integer ones are tiny."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def move(v):
        if v.is_cuda or device.type != "cuda":
            return v.to(device)
        small = not v.is_floating_point() or v.numel() <= 4096     # integer inputs, grids, second_per_grid_ts
        # The companion must be PRIVATE: version counters do not see writes through a numpy view, a raw pointer or `.data` (a collate function that
        # refills one persistent pinned buffer per batch does exactly that), and a companion that changed under the plan gives a silently wrong loss.
        # A pageable tensor is staged through a pinned copy nobody else holds: that copy is the exact source of the H2D transfer and becomes the companion;
        # a tensor the caller already pinned is the caller's, so the (small) companion is cloned from it.
        p = v if v.is_pinned() else v.pin_memory()
        d = p.to(device, non_blocking=True)
        if not small:
            return d
        return attach_host(d, p if p is not v else v.clone())

    for k, v in input_dict.items():
        if isinstance(v, torch.Tensor):
            input_dict[k] = move(v)
        elif isinstance(v, list) and len(v) > 0 and isinstance(v[0], torch.Tensor):
            input_dict[k] = [move(e) for e in v]
    return input_dict
