#!/usr/bin/env python3
"""Which leg of the multi-object stream is not reproducible?  SAM2-L at the configs[3] shape, 3 objects, 6 frames, several clips; per clip the masks of
(a) eager sequential run twice, (b) graphs replayed one object after the other, (c) graphs replayed concurrently -- compared pairwise, first mismatch located by (frame, object).
python3 tools/probes/slot_race_probe.py [clips] [variant]     variant: force<tile> (every tuned GEMM on one tiling, e.g. force20: no stream-K), norowchain (memory
attention's row steps as separate launches), serial (a device sync between the objects' replays inside a frame: concurrency off, streams on)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.model.sam2 import SAM2, MultiObjectSession, VideoSession  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 6
variant = sys.argv[2] if len(sys.argv) > 2 else ""
from rga3.hip import tuner  # noqa: E402
import rga3.model.sam2 as S2  # noqa: E402
if variant.startswith("force"):
    tuner._forced[0] = int(variant[5:])
if variant == "norowchain":
    S2._ROWCHAIN = False
if variant == "serial":
    _real_replay = torch.cuda.CUDAGraph.replay

    def _replay(self):
        torch.cuda.synchronize()
        _real_replay(self)
        torch.cuda.synchronize()
    torch.cuda.CUDAGraph.replay = _replay
if variant == "avoidws":      # tuned picks, but never a tiling that hands partial sums through the workspace (stream-K / split-K)
    _real_pick = tuner.pick

    def _pick(key, run, extra=(), candidates=None):
        t = _real_pick(key, run, extra, candidates)
        if t in (14, 21, 22, 25, 26, 27, 31, 32):
            tm = {k: v for k, v in tuner.timings().get(key, {}).items() if k not in (14, 21, 22, 25, 26, 27, 31, 32)}
            t = min(tm, key=tm.get) if tm else 20
        return t
    tuner.pick = _pick
print("variant:", variant or "(product)", flush=True)
dev = torch.device("cuda")
torch.manual_seed(23)
m = SAM2()
g = torch.Generator().manual_seed(23)
with torch.no_grad():
    for n, p in m.named_parameters():
        if p.dim() >= 2:
            p.copy_(torch.randn(p.shape, generator=g) * (0.02 if p.shape[-1] > 8 else 0.2))
        elif "norm" in n and n.endswith("weight"):
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
        else:
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
m = m.to(torch.bfloat16).to(dev).eval()
n_obj, T = 3, 6
embs = [torch.randn(1, 1, 256, generator=g).to(torch.bfloat16).to(dev) for _ in range(n_obj)]


def track(vid, feats, **kw):
    ms = MultiObjectSession(m.sam2_model, vid, n_obj, feats=feats)
    if variant == "capstreams":   # replay every slot's graphs on the stream they were captured on
        ms._streams = [S2._graph_cache(m.sam2_model, s_)[("stream", s_.slot)] for s_ in ms.sessions]
        for st_ in ms._streams:
            st_.wait_stream(torch.cuda.current_stream())
    for o in range(n_obj):
        ms.add_language_embd(0, o, embs[o])
    return torch.cat([mk for _, mk in ms.propagate(**kw)], 0).view(-1, n_obj, 1024, 1024)


def where(a, b):
    if torch.equal(a, b):
        return "equal"
    d = (a != b).flatten(2).any(2)
    idx = d.nonzero()
    return "DIFFER first (frame, obj) = %s, %d (frame, obj) pairs, max abs %.3e" % (tuple(idx[0].tolist()), idx.shape[0], float((a - b).abs().max()))


with torch.no_grad():
    for clip in range(clips):
        _shift = [torch.cuda.Stream() for _ in range(7 * clip % 32)]
        vid = torch.randn(T, 3, 1024, 1024, generator=g).to(torch.bfloat16).to(dev)
        feats = VideoSession(m.sam2_model, vid)._ensure_feats()
        e1 = track(vid, feats, use_graph=False, concurrent=False)
        torch.cuda.synchronize()
        e2 = track(vid, feats, use_graph=False, concurrent=False)
        torch.cuda.synchronize()
        gs = track(vid, feats, use_graph=True, concurrent=False)
        torch.cuda.synchronize()
        gc1 = track(vid, feats, use_graph=True, concurrent=True)
        torch.cuda.synchronize()
        gc2 = track(vid, feats, use_graph=True, concurrent=True)
        torch.cuda.synchronize()
        ec = track(vid, feats, use_graph=False, concurrent=True)
        torch.cuda.synchronize()
        print(f"clip {clip}: eager vs eager {where(e1, e2)} | graphs sequential vs eager {where(gs, e1)} | graphs concurrent vs eager {where(gc1, e1)} | "
              f"concurrent twice {where(gc1, gc2)} | eager concurrent vs eager {where(ec, e1)}", flush=True)
