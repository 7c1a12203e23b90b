#!/usr/bin/env python3
"""Where inside a tracked frame do concurrently replayed slot graphs first disagree with the same graphs replayed one object after the other?  Checkpoints inside the
captured body (memory-attention output, mask-decoder outputs, memory-encoder output; with argv[2] = layers also every memory-attention layer's output) are kept
alive, cloned on the slot's stream right after every replay (no host sync), and compared between a sequential and a concurrent run of the same clip.
python3 tools/probes/slot_race_stash.py [clips] [layers]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import rga3.model.sam2 as S2  # noqa: E402
from rga3.hip import ops  # noqa: E402
from rga3.model.sam2 import SAM2, MultiObjectSession, VideoSession  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fine = len(sys.argv) > 2 and sys.argv[2] == "layers"
dev = torch.device("cuda")
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
P = m.sam2_model
n_obj, T = 3, 6
embs = [torch.randn(1, 1, 256, generator=g).to(torch.bfloat16).to(dev) for _ in range(n_obj)]

STASH, CUR, LOG = {}, [None], []


def put(name, t):
    if CUR[0] is not None and torch.is_tensor(t):
        STASH.setdefault(CUR[0], {})[name] = t


P.memory_attention.register_forward_hook(lambda mod, i, o: put("1 memory attention out", o))
_heads, _enc = P.forward_sam_heads, P.encode_new_memory


def heads(*a, **k):
    o = _heads(*a, **k)
    put("2 low_res_masks", o["low_res_masks"]); put("2 obj_ptr", o["obj_ptr"]); put("2 high_res_masks", o["high_res_masks"])
    return o


def enc(*a, **k):
    mf, mp = _enc(*a, **k)
    put("3 memory features", mf)
    return mf, mp


P.forward_sam_heads, P.encode_new_memory = heads, enc
if fine:      # every op of the memory attention in call order (the fused path calls ops.*, not the layer modules)
    ctr = [0]
    for nm in ("add_bcast", "add", "gemm", "rope_axial_", "memlayer_rows", "attn_varlen", "memattn_cross"):
        real = getattr(ops, nm)

        def wrap(*a, _real=real, _nm=nm, **k):
            r = _real(*a, **k)
            if CUR[0] is not None and CUR[0][2] == "on":
                ctr[0] += 1
                outs = r if isinstance(r, (tuple, list)) else (r,)
                for j, t in enumerate(outs):
                    if torch.is_tensor(t):
                        put("0.%03d %s[%d]" % (ctr[0], _nm, j), t)
                if _nm == "rope_axial_":
                    put("0.%03d %s(in place)" % (ctr[0], _nm), a[0])
            return r
        setattr(ops, nm, wrap)
    _ma = P.memory_attention.forward

    def ma(*a, **k):
        ctr[0] = 0
        CUR[0] = (CUR[0][0], CUR[0][1], "on")
        try:
            return _ma(*a, **k)
        finally:
            CUR[0] = (CUR[0][0], CUR[0][1], "off")
    P.memory_attention.forward = ma

_gf = VideoSession._graph_frame


def gf(self, t, start):
    key = (self.slot, min(t - start, 99))
    CUR[0] = (key[0], key[1], "off")
    try:
        r = _gf(self, t, start)
    finally:
        CUR[0] = None
    rec = {}
    for (sl, d, _), dd in STASH.items():
        if (sl, d) == key:
            rec.update({n: v.clone() for n, v in dd.items()})      # on this slot's stream, right behind the replay
    LOG.append((t, self.slot, rec))
    return r


VideoSession._graph_frame = gf


def track(vid, feats, **kw):
    del LOG[:]
    ms = MultiObjectSession(P, vid, n_obj, feats=feats)
    for o in range(n_obj):
        ms.add_language_embd(0, o, embs[o])
    out = torch.cat([mk for _, mk in ms.propagate(**kw)], 0)
    torch.cuda.synchronize()
    return out, {(t, s): r for t, s, r in LOG}


with torch.no_grad():
    for clip in range(clips):
        vid = torch.randn(T, 3, 1024, 1024, generator=g).to(torch.bfloat16).to(dev)
        feats = VideoSession(P, vid)._ensure_feats()
        o_seq, l_seq = track(vid, feats, use_graph=True, concurrent=False)
        o_con, l_con = track(vid, feats, use_graph=True, concurrent=True)
        bad = []
        for key in sorted(l_seq):
            for name in sorted(l_seq[key]):
                a, b = l_seq[key][name], l_con[key].get(name)
                if b is not None and not torch.equal(a, b):
                    bad.append((key, name, float((a.float() - b.float()).abs().max()), int((a != b).sum()), a.numel()))
        print(f"clip {clip}: masks {'equal' if torch.equal(o_seq, o_con) else 'DIFFER'}; {len(bad)} checkpoints differ", flush=True)
        seen = set()
        for key, name, mx, cnt, numel in bad:
            if key not in seen:      # the first differing checkpoint of each (frame, slot), in execution order
                seen.add(key)
                print(f"   frame {key[0]} slot {key[1]}: first at '{name}': {cnt} of {numel} elements, max abs {mx:.3e}", flush=True)
