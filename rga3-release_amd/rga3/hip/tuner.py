"""Per-shape tile selection for the bf16 GEMM, measured on the device it runs on (MI355X boards differ by >10 % on the same
binary: MI355X_MICROARCH.md 'DVFS give-back' item 5).  First call of a (M-bucket, N, K, epilogue) shape times the candidate
tilings with HIP events on the launch stream and caches the winner for the process; RGA3_GEMM_TUNE=0 falls back to the
library's static heuristic (tile = -1)."""
from __future__ import annotations

import os

import torch

CANDIDATES = (20, 21, 22, 11, 12, 3, 4)   # 256x256 ping-pong: one tile per workgroup / persistent / persistent + stream-K tail;
                                       # single-phase 256x128, 128x128, 128x256, 128x320
_cache = {}
_times = {}   # key -> {tile: ms of 3 launches} (diagnostic, see table())
_enabled = os.environ.get("RGA3_GEMM_TUNE", "1") != "0"


def key_of(M, N, K, act, out_f32, has_bias, has_res):
    return ((M + 255) // 256, N, K, act, out_f32, has_bias, has_res)


def pick(key, run, extra=()):
    """run(tile) launches the GEMM once with that tiling.  Returns the cached / measured best tile id."""
    if not _enabled:
        return -1
    t = _cache.get(key)
    if t is not None:
        return t
    if torch.cuda.is_current_stream_capturing():
        return -1
    best, best_ms = -1, float("inf")
    _times[key] = {}
    for tile in tuple(CANDIDATES) + tuple(extra):
        run(tile)  # warm (also sets the func attribute for large dynamic LDS)
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(3):
            run(tile)
        en.record()
        en.synchronize()
        ms = st.elapsed_time(en)
        _times[key][tile] = ms / 3.0
        if ms < best_ms:
            best, best_ms = tile, ms
    _cache[key] = best
    return best


def table():
    return dict(_cache)


def timings():
    return {k: dict(v) for k, v in _times.items()}
