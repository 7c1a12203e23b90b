"""Per-shape tile selection for the bf16 GEMM, measured on the device it runs on (MI355X boards differ by >10 % on the same
binary: MI355X_MICROARCH.md 'DVFS give-back' item 5).  First call of a (M-bucket, N, K, epilogue) shape times the candidate
tilings with HIP events on the launch stream and caches the winner for the process; RGA3_GEMM_TUNE=0 falls back to the
library's static heuristic (tile = -1).

``refine(step)`` re-decides every shape IN SITU: a product measured alone re-reads operands that sit in the 256-MiB Infinity Cache
and runs at the clock of a short burst, while inside the model its weights stream from HBM and its neighbours set the clock and
the cache state -- the single-phase tilings lose 10-20 % on cold weights, the ping-pong ones do not (tools/gemm_cold.py), so the
stand-alone winner is not always the winner in place.  refine times the caller's whole step with each close candidate of each
shape and keeps what makes the STEP fastest."""
from __future__ import annotations

import os

import torch

CANDIDATES = (20, 21, 22, 27, 28, 31, 32, 12, 13, 3, 4, 5, 6)   # 256x256 ping-pong: one tile per workgroup / persistent / persistent + stream-K tail / persistent with the ragged-row loop; 28 = 256x256 on four waves;
                                               # 192x256 persistent / + stream-K tail (M = 2112 = 11 x 192); single-phase 128x128, 64x64,
                                               # 128x256, 128x320, 128x192; 128x256 with three LDS stages (two K-tiles in flight)
_cache = {}
_times = {}   # key -> {tile: ms of 3 launches} (diagnostic, see table())
_enabled = os.environ.get("RGA3_GEMM_TUNE", "1") != "0"


_forced = [None]


class force:
    """``with tuner.force(10): ...`` runs every tuned GEMM on one explicit tiling (bench.py re-runs its timed forward on the first-generation
    single-phase 256x256 kernel, id 10, and compares the outputs)."""

    def __init__(self, tile: int):
        self.tile = int(tile)

    def __enter__(self):
        self.old, _forced[0] = _forced[0], self.tile
        return self

    def __exit__(self, *exc):
        _forced[0] = self.old


def forced():
    return _forced[0]


def key_of(M, N, K, act, out_f32, has_bias, has_res):
    return ((M + 255) // 256, N, K, act, out_f32, has_bias, has_res)


def pick(key, run, extra=(), candidates=None):
    """run(tile) launches the GEMM once with that tiling.  Returns the cached / measured best tile id."""
    if _forced[0] is not None:
        return _forced[0]
    if not _enabled:
        return -1
    t = _cache.get(key)
    if t is not None:
        return t
    if torch.cuda.is_current_stream_capturing():
        return -1
    best, best_ms = -1, float("inf")
    _times[key] = {}
    for tile in tuple(CANDIDATES if candidates is None else candidates) + tuple(extra):
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


def _time_step(step, reps):
    ts = []
    for _ in range(reps):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        step()
        en.record()
        en.synchronize()
        ts.append(st.elapsed_time(en))
    ts.sort()
    return ts[len(ts) // 2]


def refine(step, reps: int = 5, within: float = 1.5, min_gain: float = 0.002, verbose=None):
    """Re-pick the tiling of every GEMM shape ``step()`` launches by the time of the whole step (median of ``reps`` runs per candidate).
    Only candidates whose stand-alone time is within ``within`` x the stand-alone best are tried; a change must win ``min_gain`` of the
    step.  Returns {key: (old tile, new tile)} for the shapes that changed.  Call it once after the first (tuning) step."""
    if not _enabled:
        return {}
    step()                                   # make sure every shape has its stand-alone table
    torch.cuda.synchronize()
    changed = {}
    base = _time_step(step, reps)
    for key in list(_cache):
        tm = _times.get(key)
        if not tm:
            continue
        cur = _cache[key]
        lim = min(tm.values()) * within
        best_t, best_ms = cur, base
        for cand in sorted((t for t in tm if t != cur and tm[t] <= lim), key=lambda t: tm[t]):
            _cache[key] = cand
            ms = _time_step(step, reps)
            if ms < best_ms * (1.0 - min_gain):
                best_t, best_ms = cand, ms
        _cache[key] = best_t
        if best_t != cur:
            changed[key] = (cur, best_t)
            base = best_ms
            if verbose:
                verbose(f"tuner.refine: {key} tile {cur} -> {best_t}: step {best_ms:.3f} ms")
    return changed


def table():
    return dict(_cache)


def save(path: str):
    """Write the decisions (key -> tile) as JSON; `load` restores them so a later process launches no trial GEMMs (rocprofv3 --pmc passes count bytes
    per launch of the TIMED tilings only)."""
    import json
    with open(path, "w") as f:
        json.dump([[list(k), v] for k, v in _cache.items()], f)


def load(path: str):
    import json
    with open(path) as f:
        for k, v in json.load(f):
            _cache[tuple(k)] = int(v)


def timings():
    return {k: dict(v) for k, v in _times.items()}


if os.environ.get("RGA3_TUNE_LOAD") and os.path.exists(os.environ["RGA3_TUNE_LOAD"]):
    load(os.environ["RGA3_TUNE_LOAD"])
