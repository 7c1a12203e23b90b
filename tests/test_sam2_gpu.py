"""GPU parity of the product SAM2 (rga3/model/sam2.py on HIP kernels, bf16) against the fp32 oracle run on the same
bf16-rounded weights, and against the golden vectors captured from the reference's own classes.
Tolerances (as stated): feature / mask-logit rel-L2 <= 2e-2, mask IoU >= 0.99, argmax-IoU index bit-exact.
The fixture clips show an object and the mask head's read-out is fitted to it (tests/blob_inputs.py, tests/golden/blobfit.py): the reference's masks
are blobs with |logit| >> 0 on > 98 % of the pixels, so the thresholded comparisons mean something."""
import numpy as np
import pytest
import torch

from oracle import sam2 as S
from tests.sam2_tiny import det_params, gold, images, lang, tiny_cfg

pytestmark = pytest.mark.gpu

TINY = dict(image_size=128, embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4), pos_bkg=(7, 7),
            d_model=256, mem_dim=64, memattn_layers=2, memattn_ff=64)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def iou(a, b):
    a, b = a.cpu().bool(), b.cpu().bool()
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u


@pytest.fixture(scope="module")
def G():
    return gold()


@pytest.fixture(scope="module")
def P(G):
    return det_params(G, bf16_round=True)


@pytest.fixture(scope="module")
def model(dev, G):
    from rga3.model.sam2 import SAM2

    m = SAM2(**TINY)
    m.sam2_model.load_state_dict(det_params(G), strict=True)   # same names / shapes as the reference state dict
    return m.to(torch.bfloat16).to(dev).eval()


def tok2map(t, Fn, H, W):
    return t.float().cpu().view(Fn, H, W, -1).permute(0, 3, 1, 2)


def test_image_encoder(model, dev, P):
    cfg = tiny_cfg()
    img = images(2).to(torch.bfloat16)
    with torch.no_grad():
        f = model.sam2_model.forward_image(img.to(dev))
        bo = S.image_encoder_forward(P, img.float(), cfg)
    assert rel(tok2map(f["feat"], 2, 8, 8), bo["backbone_fpn"][2]) < 2e-2
    assert rel(tok2map(f["feat_s1"], 2, 16, 16), bo["backbone_fpn"][1]) < 2e-2
    assert rel(tok2map(f["feat_s0"], 2, 32, 32), bo["backbone_fpn"][0]) < 2e-2
    assert rel(tok2map(f["pos"], 1, 8, 8), bo["vision_pos_enc"][2][:1]) < 5e-3


def test_train_path_masks(model, dev, P, G):
    cfg = tiny_cfg()
    img, emb = images(3).to(torch.bfloat16), lang(3).to(torch.bfloat16)
    with torch.no_grad():
        st = model.get_sam2_embeddings_train(img.to(dev))
        low, high = model.inject_language_embd_train(st, emb.to(dev))
        o = model.sam2_model.forward_sam_heads(__import__("rga3.hip.ops", fromlist=["x"]).add_bcast(st["feat"], model.sam2_model.no_mem_embed.view(1, -1)), st, emb.to(dev))
        feats = S.prepare_backbone_features(S.image_encoder_forward(P, img.float(), cfg))
        rlow, rhigh, ro = S.inject_language_embd_train(P, feats, emb.float(), cfg)
    assert low.shape == (3, 1, 32, 32) and high.shape == (3, 1, 128, 128) and high.dtype == torch.float32
    assert rel(o["low_res_multimasks"], ro["low_res_multimasks"]) < 2e-2
    assert rel(o["ious"], ro["ious"]) < 1e-2
    top2 = ro["ious"].topk(2, -1).values
    assert bool(((top2[:, 0] - top2[:, 1]) > 0.05).all())          # the fixture's IoU head is decisive ...
    assert torch.equal(o["best_iou_inds"].cpu(), ro["best_iou_inds"])   # ... so the argmax index is bit-exact on every frame
    assert np.array_equal(o["best_iou_inds"].cpu().numpy(), G["g3_heads_best"])
    assert rel(high, rhigh) < 2e-2
    assert rel(o["obj_ptr"], ro["obj_ptr"]) < 2e-2
    rh = torch.from_numpy(G["g3_train_high"])
    assert float((rh.abs() > 0.05 * rh.abs().amax((1, 2, 3), keepdim=True)).float().mean()) > 0.98   # blobs, not speckle
    for i in range(3):
        assert iou(high[i] > 0, rhigh[i] > 0) >= 0.99
        assert iou(high[i] > 0, rh[i] > 0) >= 0.99  # vs the reference's own fp32 masks (unrounded weights)


def test_inference_prompt_every_frame(model, dev, P, G):
    cfg = tiny_cfg()
    img, emb = images().to(torch.bfloat16), lang().to(torch.bfloat16)
    with torch.no_grad():
        sess = model.get_sam2_embeddings(img.to(dev))
        masks = model.language_embd_inference(sess, [emb[t].to(dev) for t in range(5)])
        rmasks, rs = S.language_embd_inference(P, img.float(), [emb[t].float() for t in range(5)], cfg)
    assert masks.shape == (5, 1, 128, 128)
    assert sess.counts["enc"] == 5 and sess.counts["memattn"] == 0 and sess.counts["memenc"] == 0   # reference: enc=10, memenc=5 (dead work)
    same = [int(sess.cond[t]["best_iou_inds"]) == int(rs.out["cond_frame_outputs"][t]["best_iou_inds"]) if "best_iou_inds" in rs.out["cond_frame_outputs"][t] else True for t in range(5)]
    assert all(same)
    ious = [iou(masks[t] > 0, rmasks[t] > 0) for t in range(5)]
    margin = rmasks.abs() > 0.05 * rmasks.abs().max()
    assert margin.float().mean() > 0.98                                       # blobs with a real margin
    assert torch.equal((masks.cpu() > 0)[margin], (rmasks > 0)[margin])       # bit-exact outside the band at the blob edges
    assert rel(masks, rmasks) < 2e-2
    assert min(ious) >= 0.99, ious
    gm = torch.from_numpy(G["g3_infer_all_masks"])                            # the reference's own output (fp32, unrounded weights, 2x encoder passes)
    assert min(iou(masks[t] > 0, gm[t] > 0) for t in range(5)) >= 0.99


def test_frame0_prompt_propagation(model, dev, P, G):
    from rga3.model.sam2 import VideoSession

    cfg = tiny_cfg()
    img, emb = images().to(torch.bfloat16), lang().to(torch.bfloat16)
    with torch.no_grad():
        sess = VideoSession(model.sam2_model, img.to(dev))
        sess.add_language_embd(0, emb[0][None].to(dev))
        res = sess.propagate()
        rs = S.VideoSession(P, img.float(), cfg)
        rs.add_language_embd(0, emb[0][None].float())
        rres = rs.propagate()
    assert sess.counts["memattn"] == 4 and sess.counts["enc"] == 5
    masks, rmasks = torch.cat([m for _, m in res]), torch.cat([m for _, m in rres])
    assert rel(masks, rmasks) < 2e-2
    assert min(iou(masks[t] > 0, rmasks[t] > 0) for t in range(5)) >= 0.99
    gm = torch.from_numpy(G["g3_prop0_masks"])                                # the reference's own propagate_in_video output
    assert float((gm.abs() > 0.05 * gm.abs().max()).float().mean()) > 0.98
    assert min(iou(masks[t] > 0, gm[t] > 0) for t in range(5)) >= 0.99
    ptr = torch.stack([(sess.cond if t == 0 else sess.non_cond)[t]["obj_ptr"].float().cpu().reshape(-1) for t in range(5)])
    assert rel(ptr, torch.from_numpy(G["g3_prop0_obj_ptrs"]).reshape(5, -1)) < 2e-2


def test_reverse_and_ranged_propagation(model, dev, P):
    """VideoSession.propagate(start_frame_idx, max_frame_num_to_track, reverse) against the reference's own propagate_in_video outputs (tests/golden/
    sam2_reverse.npz) and the oracle: processing order exact; masks rel-L2 <= 2e-2 / IoU >= 0.99; object pointers <= 2e-2.  Case A runs a forward pass and then a
    reverse pass on the same session: the reverse pass reads the memories the forward pass left on frames 3 and 4 -- one of which the forward pass never encoded
    (nothing after frame 4 read it), so it is encoded on first use."""
    import os
    from rga3.model.sam2 import VideoSession

    R = np.load(os.path.join(os.path.dirname(__file__), "golden", "sam2_reverse.npz"))
    cfg = tiny_cfg()
    img, emb = images().to(torch.bfloat16), lang().to(torch.bfloat16)

    def run(prompt, passes):
        with torch.no_grad():
            sess = VideoSession(model.sam2_model, img.to(dev))
            sess.add_language_embd(prompt, emb[0][None].to(dev))
            return sess, [sess.propagate(**kw) for kw in passes]

    def check(res, frames, gm):
        assert [t for t, _ in res] == frames.tolist()
        masks, gm = torch.cat([m for _, m in res]), torch.from_numpy(gm)
        assert rel(masks, gm) < 2e-2, rel(masks, gm)
        assert min(iou(masks[i] > 0, gm[i] > 0) for i in range(len(frames))) >= 0.99

    sess, (fa, ra) = run(2, [dict(), dict(start_frame_idx=2, reverse=True)])
    check(fa, R["A_fwd_frames"], R["A_fwd_masks"])
    check(ra, R["A_rev_frames"], R["A_rev_masks"])
    ptr = torch.stack([(sess.cond.get(t) or sess.non_cond[t])["obj_ptr"].float().cpu().reshape(-1) for t in range(5)])
    assert rel(ptr, torch.from_numpy(R["A_obj_ptrs"]).reshape(5, -1)) < 2e-2
    assert sess.counts["enc"] == 5      # every frame encoded once over both passes
    _, (rb,) = run(4, [dict(reverse=True, max_frame_num_to_track=2)])
    check(rb, R["B_frames"], R["B_masks"])
    _, (rc,) = run(1, [dict(start_frame_idx=1, max_frame_num_to_track=2)])
    check(rc, R["C_frames"], R["C_masks"])
    _, (rd,) = run(0, [dict(reverse=True)])
    assert rd == [] and R["D_frames"].size == 0
    with pytest.raises(RuntimeError):
        VideoSession(model.sam2_model, img.to(dev)).propagate()


def test_graph_replay_equals_eager_stream(model, dev):
    """A 24-frame stream prompted on frame 0: every later frame runs as the replay of a captured hipGraph over static buffers (one graph per
    bank state: 15 growing states, then the steady one); masks, pointers and memories must equal the eager frame-by-frame path bit for bit
    (same kernels, same inputs), also for a second session that reuses the graphs kept on the model."""
    from rga3.model.sam2 import VideoSession

    torch.manual_seed(5)
    T = 24
    vid = (torch.randn(T, 3, 128, 128) * 0.5).to(torch.bfloat16).to(dev)
    emb = torch.randn(1, 1, 256).to(torch.bfloat16).to(dev)
    with torch.no_grad():
        a = VideoSession(model.sam2_model, vid)
        a.add_language_embd(0, emb)
        ra = a.propagate()
        b = VideoSession(model.sam2_model, vid, feats=a.feats)
        b.add_language_embd(0, emb)
        rb = b.propagate(use_graph=True)
        c = VideoSession(model.sam2_model, vid, feats=a.feats)     # a second session captures its own graph: nothing is shared between them
        c.add_language_embd(0, emb)
        rc = c.propagate(use_graph=True)
    assert len(ra) == len(rb) == len(rc) == T and b.counts["memattn"] == a.counts["memattn"] == T - 1
    for (ta, ma), (tb, mb), (tc, mc) in zip(ra, rb, rc):
        assert ta == tb == tc and torch.equal(ma, mb) and torch.equal(ma, mc), ta
    for t in range(1, T - 1):
        assert torch.equal(a.non_cond[t]["obj_ptr"], b.non_cond[t]["obj_ptr"]) and torch.equal(a.non_cond[t]["obj_ptr"], c.non_cond[t]["obj_ptr"])
        assert torch.equal(a.non_cond[t]["maskmem_features"], b.non_cond[t]["maskmem_features"])
    assert sum(1 for k in model.sam2_model._frame_graphs if isinstance(k, tuple) and isinstance(k[0], int)) == 16   # one graph per bank state


def test_prompt_every_frame_graph_equals_eager(model, dev):
    """language_embd_inference (the evaluate() path: language prompt on every frame) replays one captured hipGraph per frame; masks must equal
    the eager per-frame path bit for bit, for a second clip as well (the graph is kept on the model)."""
    from rga3.model.sam2 import VideoSession

    torch.manual_seed(9)
    for rep in range(2):
        T = 6
        vid = (torch.randn(T, 3, 128, 128) * 0.5).to(torch.bfloat16).to(dev)
        embs = [[torch.randn(1, 256).to(torch.bfloat16).to(dev)] for _ in range(T)]
        with torch.no_grad():
            a = VideoSession(model.sam2_model, vid)
            for t in range(T):
                a.add_language_embd(t, embs[t][0].reshape(1, 1, -1))
            ra = torch.cat([mk for _, mk in a.propagate()], dim=0)
            rb = model.language_embd_inference(VideoSession(model.sam2_model, vid, feats=a.feats), embs)
        assert rb.shape[0] == T and torch.equal(ra, rb.reshape(ra.shape))


def test_frame_graphs_dropped_when_weights_change(model, dev):
    """The captured prompt step reads weights through pointers: after an in-place weight update the graphs must be re-captured, and the result
    must equal the eager path on the NEW weights."""
    from rga3.model.sam2 import VideoSession

    torch.manual_seed(13)
    vid = (torch.randn(3, 3, 128, 128) * 0.5).to(torch.bfloat16).to(dev)
    embs = [[torch.randn(1, 256).to(torch.bfloat16).to(dev)] for _ in range(3)]
    w = model.sam2_model.sam_mask_decoder.iou_prediction_head.layers[0].weight
    with torch.no_grad():
        s0 = VideoSession(model.sam2_model, vid)
        r0 = model.language_embd_inference(s0, embs)
        keep = w.clone()
        try:
            w.mul_(1.5)
            for p in model.sam2_model.sam_mask_decoder.output_hypernetworks_mlps.parameters():
                p.mul_(0.5)
            a = VideoSession(model.sam2_model, vid, feats=s0.feats)
            for t in range(3):
                a.add_language_embd(t, embs[t][0].reshape(1, 1, -1))
            ra = torch.cat([mk for _, mk in a.propagate()], dim=0)
            rb = model.language_embd_inference(VideoSession(model.sam2_model, vid, feats=s0.feats), embs)
            assert torch.equal(ra, rb.reshape(ra.shape)) and not torch.equal(rb, r0)
        finally:
            w.copy_(keep)
            for p in model.sam2_model.sam_mask_decoder.output_hypernetworks_mlps.parameters():
                p.mul_(2.0)


def test_lora_on_sam2_projections_train_equals_eval_and_gets_gradients(dev, G):
    """the reference's LoRA target filter (train_joint.py:199-212, missing comma) also wraps q_proj / v_proj of SAM2's mask decoder.
    With non-zero lora_B the autograd (training) forward must equal the fused no_grad forward, and lora_A / lora_B must receive gradients."""
    from rga3.model.qwen_train import LoRALinear, add_lora
    from rga3.model.sam2 import SAM2

    m = SAM2(**TINY)
    m.sam2_model.load_state_dict(det_params(G), strict=True)
    m = m.to(torch.bfloat16).to(dev)
    hits = add_lora(m, r=8, alpha=16, dropout=0.0)      # default exclude = the reference's (buggy) list
    dec_hits = [h for h in hits if "sam_mask_decoder" in h]
    assert dec_hits and all(h.endswith(("q_proj", "v_proj")) for h in hits)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lora_B" in n:
                p.normal_(0, 0.05)
    m.eval()
    img, emb = images(2).to(torch.bfloat16).to(dev), lang(2).to(torch.bfloat16).to(dev)
    with torch.no_grad():
        st = m.get_sam2_embeddings_train(img)
        low_e, high_e = m.inject_language_embd_train(st, emb)
        # the LoRA update must matter for this check to mean anything
        for n, p in m.named_parameters():
            if "lora_B" in n:
                p.mul_(0.0)
        low_0, _ = m.inject_language_embd_train(st, emb)
        torch.manual_seed(0)
        for n, p in m.named_parameters():
            if "lora_B" in n:
                p.normal_(0, 0.05)
        low_e, high_e = m.inject_language_embd_train(st, emb)
    assert rel(low_e, low_0) > 1e-3
    for n, p in m.named_parameters():
        p.requires_grad_("lora_" in n or "sam_mask_decoder" in n)
    with torch.enable_grad():
        low_t, high_t = m.inject_language_embd_train(st, emb)
        high_t.float().square().mean().backward()
    assert rel(low_t.detach(), low_e) < 1e-2
    got = {n: p.grad for n, p in m.named_parameters() if "lora_" in n and "sam_mask_decoder" in n}
    assert got and all(g is not None and torch.isfinite(g.float()).all() for g in got.values())
    assert any(float(g.float().abs().max()) > 0 for n, g in got.items() if "lora_A" in n)
    assert any(float(g.float().abs().max()) > 0 for n, g in got.items() if "lora_B" in n)
    assert isinstance(m.sam2_model.sam_mask_decoder.transformer.layers[0].self_attn.q_proj, LoRALinear)


def test_two_objects_against_the_reference(model, dev, P):
    """n_obj = 2 (VERDICT r4 missing items 2 and 4) against the reference's OWN outputs, tests/golden/sam2_multiobj.npz (made by make_sam2_multiobj_fixtures.py from
    /root/reference/model/sam2.py:378-404, :3824-4132): (A) language_embd_inference with two prompts per frame -> [T * n_obj, 1, S, S] FRAME-major; (B) two objects prompted
    on frame 0 and tracked together on shared image features (MultiObjectSession: one stream and one set of frame graphs per object) -> per-frame [n_obj, 1, S, S].
    Stated tolerances: mask rel-L2 <= 2e-2, IoU >= 0.99, sign equal outside the 5 % band at the blob edges; swapping the two objects must fail; the concurrent-stream
    graph replay equals the sequential eager run bit for bit."""
    from rga3.model.sam2 import MultiObjectSession
    from tests.sam2_tiny import gold_multiobj, lang2

    R = gold_multiobj()
    img, e0, e1 = images().to(torch.bfloat16).to(dev), lang().to(torch.bfloat16).to(dev), lang2().to(torch.bfloat16).to(dev)
    ga, gb = torch.from_numpy(R["A_masks"]), torch.from_numpy(R["B_masks"])
    with torch.no_grad():
        sess = model.get_sam2_embeddings(img)
        ma = model.language_embd_inference(sess, [torch.cat([e0[t], e1[t]], 0) for t in range(5)])
    assert tuple(ma.shape) == (10, 1, 128, 128)
    assert rel(ma, ga) < 2e-2 and min(iou(ma[i] > 0, ga[i] > 0) for i in range(10)) >= 0.99
    band = ga.abs() > 0.05 * ga.abs().max()
    assert torch.equal((ma.cpu() > 0)[band], (ga > 0)[band])
    assert rel(ma.reshape(5, 2, 1, 128, 128).flip(1).reshape(10, 1, 128, 128), ga) > 5e-2          # object order matters in the fixture
    assert sess.counts["enc"] == 5                                                                  # one encoder pass per frame serves both objects (reference: 10)

    def track(**kw):
        ms = MultiObjectSession(model.sam2_model, img, 2, feats=sess._ensure_feats())
        ms.add_language_embd(0, 0, e0[0][None])
        ms.add_language_embd(0, 1, e1[0][None])
        res = ms.propagate(**kw)
        return ms, res

    with torch.no_grad():
        ms, res = track(use_graph=True, concurrent=True)
        _, res_seq = track(use_graph=False, concurrent=False)
    torch.cuda.synchronize()
    assert [t for t, _ in res] == list(range(5)) and all(tuple(m.shape) == (2, 1, 128, 128) for _, m in res)
    mb = torch.cat([m for _, m in res], 0)
    assert torch.equal(mb, torch.cat([m for _, m in res_seq], 0))                                   # streams + graphs change nothing
    assert rel(mb, gb) < 2e-2 and min(iou(mb[i] > 0, gb[i] > 0) for i in range(10)) >= 0.99
    ptr = torch.stack([torch.cat([(s.cond if t == 0 else s.non_cond)[t]["obj_ptr"].float().cpu().reshape(1, -1) for s in ms.sessions], 0) for t in range(5)])
    assert rel(ptr, torch.from_numpy(R["B_obj_ptrs"])) < 2e-2
    assert ms.sessions[0].counts["memattn"] == ms.sessions[1].counts["memattn"] == int(R["B_counts"][1])     # one memory-attention pass per tracked frame and object


def test_concurrent_replay_of_captured_slot_graphs_is_bit_exact(model, dev):
    """the frame graphs of the object slots replay CONCURRENTLY on one stream per object; every slot's capture must therefore own its scratch
    (stream-K slabs / flags, the memory cross-attention's partial sums, TN counters -- ops.py keys them by (device, stream), and a capture bakes the pointers in).
    The first tracked clip captures (each capture starts with a device sync: nothing overlaps), so the check is on the SECOND clip onward: four slots replaying
    already-captured graphs side by side, several clips in a row, bit for bit against the sequential eager run of the same objects."""
    from rga3.hip import ops
    from rga3.model.sam2 import MultiObjectSession, _graph_cache

    n_obj = 4
    g = torch.Generator().manual_seed(11)
    embs = [(torch.randn(1, 1, 256, generator=g) * 0.5).to(torch.bfloat16).to(dev) for _ in range(n_obj)]

    def track(img, feats, **kw):
        ms = MultiObjectSession(model.sam2_model, img, n_obj, feats=feats)
        for o in range(n_obj):
            ms.add_language_embd(0, o, embs[o], use_graph=kw.get("use_graph", False))
        return ms, torch.cat([m for _, m in ms.propagate(**kw)], 0)

    with torch.no_grad():
        for clip in range(12):
            # torch.cuda.Stream() cycles through a pool of 32 handles: shift the cycle so that, over the clips, the objects' eager streams coincide with every
            # slot's CAPTURE stream in turn -- with stream-keyed scratch an eager launch then shared a replaying graph's scratch (the round-6 intermittent mismatch)
            _shift = [torch.cuda.Stream() for _ in range(3 * clip % 32)]
            img = images(5).roll(clip % 5, 0).to(torch.bfloat16).to(dev)
            feats = model.get_sam2_embeddings(img)._ensure_feats()
            ms, got = track(img, feats, use_graph=True, concurrent=True)
            _, want = track(img, feats, use_graph=False, concurrent=False)
            torch.cuda.synchronize()
            assert torch.equal(got, want), f"clip {clip}: concurrent graph replay differs from the sequential eager run"
    # distinct objects really produced distinct masks (the comparison above is not between constants)
    assert not torch.equal(got[0], got[1])
    cache = _graph_cache(model.sam2_model)
    scopes = {("scope", cache[("scope", o)]) for o in range(n_obj)}
    assert len(scopes) == n_obj                                        # one scratch scope per slot ...
    keyed = {k[1] for k in list(ops._gemm_ws) + list(ops._memattn_ws) + list(ops._tn_cnt) if k[1] in scopes}
    assert len(keyed) >= 2, (keyed, scopes)                            # ... and the graphs' scratch is keyed by it, not by a stream handle
