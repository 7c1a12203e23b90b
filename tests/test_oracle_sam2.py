"""CPU: pins the SAM2 oracle (oracle/sam2.py) against golden vectors produced by the reference's own model/sam2.py
classes (tests/golden/make_sam2_fixtures.py).  fp32 vs fp32: <= 1e-4 abs on O(1) tensors; integer outputs bit-exact."""
import numpy as np
import pytest
import torch

from oracle import sam2 as S
from oracle.detweights import det_tensor
from tests.sam2_tiny import det_params, gold, images, lang, tiny_cfg


@pytest.fixture(scope="module")
def G():
    return gold()


@pytest.fixture(scope="module")
def P(G):
    return det_params(G)


def close(a, b, tol=1e-4):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_g1_pure_functions(G):
    cos, sin = S.compute_axial_cis(32, 4, 4)
    assert close(cos, G["g1_axial_cos"], 1e-6) and close(sin, G["g1_axial_sin"], 1e-6)
    xq, xk = det_tensor("g1_xq", (1, 1, 16, 32)), det_tensor("g1_xk", (1, 1, 37, 32))
    q, k = S.apply_rotary_enc(xq, xk[:, :, :32], cos, sin, repeat_freqs_k=True)
    assert close(q, G["g1_rot_q"], 1e-6) and close(k, G["g1_rot_k"], 1e-6)
    xw = det_tensor("g1_win", (2, 10, 13, 6))
    win, pad = S.window_partition(xw, 4)
    assert np.array_equal(win.numpy(), G["g1_win_part"]) and tuple(pad) == tuple(G["g1_win_pad"])
    assert np.array_equal(S.window_unpartition(win, 4, pad, (10, 13)).numpy(), G["g1_win_unpart"])
    assert close(S.position_embedding_sine(32, 6, 7), G["g1_pe_sine"], 1e-6)
    assert close(S.position_embedding_random(det_tensor("g1_gauss", (2, 16)), 5, 7), G["g1_pe_random"], 1e-5)
    assert close(S.get_1d_sine_pe(torch.tensor([0.0, 0.25, 1.0]), 16), G["g1_1d_sine"], 1e-6)
    sel, unsel = S.select_closest_cond_frames(10, {t: t for t in (0, 3, 9, 14, 20)}, 3)
    assert sorted(sel) == G["g1_sel_cond"].tolist() and sorted(unsel) == G["g1_unsel_cond"].tolist()
    Pn = {"n.weight": det_tensor("g1_ln2w", (6,), 0.5, offset=1.0), "n.bias": det_tensor("g1_ln2b", (6,), 0.2)}
    assert close(S.layer_norm_2d(det_tensor("g1_ln2x", (2, 6, 3, 4)), Pn, "n"), G["g1_ln2d"], 1e-5)


def test_block_table_matches_reference_layout():
    t = S.Sam2Cfg().block_table()  # SAM2-L
    assert [r["window"] for r in t[:3]] == [8, 8, 8] and t[3]["window"] == 4 and t[9]["window"] == 16 and t[45]["window"] == 8
    assert [i for i, r in enumerate(t) if r["window"] == 0] == [23, 33, 43]
    assert [i for i, r in enumerate(t) if r["pool"]] == [2, 8, 44]
    assert [t[i]["dim_out"] for i in (0, 2, 8, 44)] == [144, 288, 576, 1152] and [t[i]["heads"] for i in (0, 2, 8, 44)] == [2, 4, 8, 16]
    assert S.Sam2Cfg().channel_list == [1152, 576, 288, 144]


def test_g2_image_encoder(G, P):
    cfg = tiny_cfg()
    with torch.no_grad():
        tr = S.hiera_forward(P, images(1), cfg)
        for i, f in enumerate(tr):
            assert close(f, G[f"g2_trunk_{i}"]), i
        bo = S.image_encoder_forward(P, images(2), cfg)
    for i in range(3):
        assert close(bo["backbone_fpn"][i], G[f"g2_fpn_{i}"]), i
        assert close(bo["vision_pos_enc"][i][0], G[f"g2_pos_{i}"], 1e-5), i


def test_g3_train_path_and_heads(G, P):
    cfg = tiny_cfg()
    with torch.no_grad():
        feats = S.prepare_backbone_features(S.image_encoder_forward(P, images(3), cfg))
        low, high, o = S.inject_language_embd_train(P, feats, lang(3), cfg)
        assert np.array_equal(o["best_iou_inds"].numpy(), G["g3_heads_best"])  # argmax index: bit-exact
        assert close(o["ious"], G["g3_heads_ious"], 1e-5)
        assert close(o["low_res_multimasks"], G["g3_heads_low_multi"])
        assert close(o["obj_ptr"], G["g3_heads_obj_ptr"]) and close(o["object_score_logits"], G["g3_heads_obj_logits"])
        assert close(low, G["g3_train_low"]) and close(high, G["g3_train_high"])
        mf, mp = S.encode_new_memory(P, feats[0], feats[2], o["high_res_masks"], cfg)
        assert close(mf, G["g2_memenc_feat"]) and close(mp, G["g2_memenc_pos"], 1e-5)
        mem, mpos = det_tensor("g2_mem", (136, 1, 64)), det_tensor("g2_mem_pos", (136, 1, 64))
        ma = S.memory_attention(P, feats[0][-1][:, :1], feats[1][-1][:, :1], mem, mpos, 8, cfg)
        assert close(ma, G["g2_memattn"])


def test_g3_inference_prompt_every_frame(G, P):
    cfg = tiny_cfg()
    with torch.no_grad():
        masks, sess = S.language_embd_inference(P, images(), [lang()[t] for t in range(5)], cfg)
    assert close(masks, G["g3_infer_all_masks"])
    assert [sess.counts[k] for k in ("enc", "memattn", "memenc", "dec")] == G["g3_infer_all_counts"].tolist()  # enc=2T, memattn=0
    assert np.array_equal((masks > 0).numpy(), G["g3_infer_all_masks"] > 0)


def test_g3_frame0_prompt_propagation_uses_memory_attention(G, P):
    cfg = tiny_cfg()
    with torch.no_grad():
        sess = S.VideoSession(P, images(), cfg)
        sess.add_language_embd(0, lang()[0][None])
        res = sess.propagate()
    masks = torch.cat([m for _, m in res], 0)
    assert [sess.counts[k] for k in ("enc", "memattn", "memenc", "dec")] == G["g3_prop0_counts"].tolist()
    assert sess.counts["memattn"] == 4
    ptrs = np.stack([(sess.out["cond_frame_outputs"] if t == 0 else sess.out["non_cond_frame_outputs"])[t]["obj_ptr"].numpy() for t in range(5)])
    assert close(ptrs, G["g3_prop0_obj_ptrs"], 2e-4)
    assert close(masks, G["g3_prop0_masks"], 2e-4)


def test_g3_reverse_and_ranged_propagation(P):
    """propagate_in_video(start_frame_idx, max_frame_num_to_track, reverse) and the track_in_reverse memory selection, against the reference's own outputs
    (tests/golden/sam2_reverse.npz, made by make_sam2_reverse_fixtures.py): processing order exact, masks and object pointers <= 2e-4."""
    import os
    R = np.load(os.path.join(os.path.dirname(__file__), "golden", "sam2_reverse.npz"))
    cfg = tiny_cfg()
    img, emb = images(), lang()

    def run(prompt, passes):
        sess = S.VideoSession(P, img, cfg)
        with torch.no_grad():
            sess.add_language_embd(prompt, emb[0][None])
            return sess, [sess.propagate(**kw) for kw in passes]

    sess, (fa, ra) = run(2, [dict(), dict(start_frame_idx=2, reverse=True)])
    assert [t for t, _ in fa] == R["A_fwd_frames"].tolist() and [t for t, _ in ra] == R["A_rev_frames"].tolist()
    assert close(torch.cat([m for _, m in fa]), R["A_fwd_masks"], 2e-4) and close(torch.cat([m for _, m in ra]), R["A_rev_masks"], 2e-4)
    od = sess.out
    ptrs = torch.stack([(od["cond_frame_outputs"].get(t) or od["non_cond_frame_outputs"][t])["obj_ptr"] for t in range(5)])
    assert close(ptrs, R["A_obj_ptrs"], 2e-4)
    _, (rb,) = run(4, [dict(reverse=True, max_frame_num_to_track=2)])
    assert [t for t, _ in rb] == R["B_frames"].tolist() and close(torch.cat([m for _, m in rb]), R["B_masks"], 2e-4)
    _, (rc,) = run(1, [dict(start_frame_idx=1, max_frame_num_to_track=2)])
    assert [t for t, _ in rc] == R["C_frames"].tolist() and close(torch.cat([m for _, m in rc]), R["C_masks"], 2e-4)
    _, (rd,) = run(0, [dict(reverse=True)])
    assert rd == [] and R["D_frames"].size == 0


def test_g3_two_objects_layout_and_values(P):
    """n_obj = 2 against the reference's own outputs (tests/golden/sam2_multiobj.npz, made by make_sam2_multiobj_fixtures.py): (A) SAM2.language_embd_inference with two
    prompts per frame returns [T * n_obj, 1, S, S], FRAME-major (reference sam2.py:378-404); (B) two objects prompted on frame 0 and tracked as one batch
    (:3977-4132): per-frame yields [n_obj, 1, S, S], object pointers per object.  Values <= 2e-4, binarised masks equal, and swapping the objects must NOT pass."""
    from tests.sam2_tiny import gold_multiobj, lang2
    R = gold_multiobj()
    cfg = tiny_cfg()
    img, e0, e1 = images(), lang(), lang2()
    with torch.no_grad():
        masks, _ = S.language_embd_inference(P, img, [torch.cat([e0[t], e1[t]], 0) for t in range(5)], cfg)
    assert tuple(masks.shape) == tuple(R["A_masks"].shape) == (10, 1, 128, 128)
    assert close(masks, R["A_masks"], 2e-4) and np.array_equal((masks > 0).numpy(), R["A_masks"] > 0)
    swapped = masks.reshape(5, 2, 1, 128, 128).flip(1).reshape(10, 1, 128, 128)
    assert not close(swapped, R["A_masks"], 2e-2)                       # the fixture tells the objects apart
    with torch.no_grad():
        sess = S.MultiObjectSession(P, img, cfg, 2)
        sess.add_language_embd(0, 0, e0[0][None])
        sess.add_language_embd(0, 1, e1[0][None])
        res = sess.propagate()
    assert [t for t, _ in res] == list(range(5)) and all(tuple(m.shape) == (2, 1, 128, 128) for _, m in res)
    mb = torch.cat([m for _, m in res], 0)
    assert close(mb, R["B_masks"], 2e-4) and np.array_equal((mb > 0).numpy(), R["B_masks"] > 0)
    ptrs = torch.stack([torch.cat([(s.out["cond_frame_outputs"] if t == 0 else s.out["non_cond_frame_outputs"])[t]["obj_ptr"] for s in sess.sessions], 0) for t in range(5)])
    assert close(ptrs, R["B_obj_ptrs"], 2e-4)
    # what the batch costs the reference per frame (one batched call each) is what ONE restated object session counts
    assert [sess.sessions[0].counts[k] for k in ("memattn", "memenc")] == R["B_counts"][1:3].tolist()
