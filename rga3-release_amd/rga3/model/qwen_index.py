"""Host-side integer plumbing of Qwen2.5-VL (bit-exact targets).

Vectorised numpy; runs on the host once per batch (the reference does the same work in Python loops:
transformers/vision_utils.py:42-188 and modeling_qwen2_5_vl.py:944-1058 of the HF module that
reference model/qwen_2_5_vl_sam2.py:104 subclasses).
"""
from __future__ import annotations

import numpy as np


def vision_cu_seqlens(grid_thw) -> np.ndarray:
    """One full-attention segment per temporal slice: cumulative h*w, int32 (vision_utils.py:42-65)."""
    g = np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3)
    seg = np.repeat(g[:, 1] * g[:, 2], g[:, 0])
    return np.concatenate([[0], np.cumsum(seg)]).astype(np.int32)


def vision_position_ids(grid_thw, merge: int) -> np.ndarray:
    """(h, w) rotary ids, block-major over merge x merge blocks, tiled over t (vision_utils.py:81-127)."""
    out = []
    for t, h, w in np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3).tolist():
        hh, ww = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        shape = (h // merge, merge, w // merge, merge)
        hh = hh.reshape(shape).transpose(0, 2, 1, 3).reshape(-1)
        ww = ww.reshape(shape).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hh, ww], -1), (t, 1)))
    return np.concatenate(out, 0).astype(np.int64)


def vision_window_index(grid_thw, merge: int, window_size: int, patch_size: int):
    """Window permutation (merge-unit granularity) and cumulative window lengths in patches
    (vision_utils.py:130-188, including the always-added pad window that unique_consecutive removes)."""
    ws = window_size // merge // patch_size
    unit = merge * merge
    idx_all, cu, base = [], [np.zeros(1, dtype=np.int64)], 0
    last = 0
    for t, h, w in np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3).tolist():
        lh, lw = h // merge, w // merge
        ph, pw = ws - lh % ws, ws - lw % ws
        nh, nw = (lh + ph) // ws, (lw + pw) // ws
        idx = np.full((t, lh + ph, lw + pw), -1, dtype=np.int64)
        idx[:, :lh, :lw] = np.arange(t * lh * lw).reshape(t, lh, lw)
        idx = idx.reshape(t, nh, ws, nw, ws).transpose(0, 1, 3, 2, 4).reshape(t * nh * nw, ws * ws)
        lens = (idx >= 0).sum(1)
        idx_all.append(idx[idx >= 0] + base)
        c = np.cumsum(lens) * unit + last
        cu.append(c)
        last = int(c[-1])
        base += t * lh * lw
    cu = np.concatenate(cu)
    keep = np.concatenate([[True], cu[1:] != cu[:-1]])
    return np.concatenate(idx_all).astype(np.int64), cu[keep].astype(np.int32)


def rope_index(input_ids, image_token_id: int, video_token_id: int, merge: int, tokens_per_second: int,
               image_grid_thw=None, video_grid_thw=None, second_per_grid_ts=None, attention_mask=None,
               temporal_rule: str = "hf449"):
    """3-axis mRoPE position ids [3, B, S] (+ rope deltas [B, 1]).

    temporal_rule "hf449" restates the release the reference pins (transformers 4.49.0.dev0: temporal index
    = floor(t * second_per_grid_t * tokens_per_second), next text position = max id + 1); "hf515" restates the
    installed 5.15 module (modeling_qwen2_5_vl.py:1016-1040).  They coincide for integer second_per_grid_ts
    whenever the temporal extent does not exceed the spatial one.
    """
    ids = np.asarray(input_ids)
    B, S = ids.shape
    pos = np.zeros((3, B, S), dtype=np.int64)
    deltas = np.zeros((B, 1), dtype=np.int64)
    imgs = [] if image_grid_thw is None else np.asarray(image_grid_thw, dtype=np.int64).reshape(-1, 3).tolist()
    vids = [] if video_grid_thw is None else np.asarray(video_grid_thw, dtype=np.int64).reshape(-1, 3).tolist()
    secs = None if second_per_grid_ts is None else np.asarray(second_per_grid_ts, dtype=np.float64).reshape(-1).tolist()
    ii = vi = 0
    for b in range(B):
        keep = np.ones(S, dtype=bool) if attention_mask is None else np.asarray(attention_mask)[b].astype(bool)
        cur = ids[b][keep]
        kind = (cur == image_token_id).astype(np.int8) + 2 * (cur == video_token_id).astype(np.int8)
        # run boundaries of equal token kind
        edges = np.flatnonzero(np.diff(kind)) + 1
        starts = np.concatenate([[0], edges])
        ends = np.concatenate([edges, [len(cur)]])
        out = np.zeros((3, len(cur)), dtype=np.int64)
        p = 0
        for s, e in zip(starts.tolist(), ends.tolist()):
            if e == s:
                continue
            k = int(kind[s])
            if k == 0:
                out[:, s:e] = np.arange(e - s) + p
                p += e - s
                continue
            while s < e:  # a run may contain several grids only if nothing separates them
                if k == 1:
                    t, h, w = imgs[ii]; ii += 1
                    spg = None
                else:
                    t, h, w = vids[vi]
                    spg = 1.0 if secs is None else secs[vi]
                    vi += 1
                lh, lw = h // merge, w // merge
                n = t * lh * lw
                tt = np.arange(t, dtype=np.int64)
                if k == 2:
                    if temporal_rule == "hf515":
                        tt = tt * (tokens_per_second * int(spg))
                    else:
                        tt = np.floor(tt * spg * tokens_per_second).astype(np.int64)
                out[0, s:s + n] = np.repeat(tt, lh * lw) + p
                out[1, s:s + n] = np.tile(np.repeat(np.arange(lh), lw), t) + p
                out[2, s:s + n] = np.tile(np.arange(lw), t * lh) + p
                p = p + max(lh, lw) if temporal_rule == "hf515" else int(out[:, s:s + n].max()) + 1
                s += n
        pos[:, b, keep] = out
        deltas[b, 0] = out.max() + 1 - len(cur) if len(cur) else 0
    return pos, deltas
