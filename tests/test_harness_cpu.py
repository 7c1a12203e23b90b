"""CPU: host-side harness logic of the product (label masking, metrics, STOM warp) against the oracle loop restatements
and the golden values captured from the reference's own helpers."""
import numpy as np
import torch

from oracle import harness as H
from rga3.model import STOM as ST
from rga3.utils import data as D
from rga3.utils import metrics as M
from tests.sam2_tiny import gold


def test_label_masking_rule():
    S, E, U, A, PAD = 900, 901, 902, 903, 0
    seqs = [
        [S, 5, 6, E, S, U, 7, 8, 9, E, S, A, 10, 11, 12, 13, E, 14, PAD, PAD],
        [S, 5, E, S, U, 7, E, S, A, 10, 20, E, S, U, 3, E, S, A, 10, 30],
        [PAD, PAD, S, 1, E, S, A, 10, E, S, A, 10, 77, 88, E, 4, 4, 4, 4, 4],
    ]
    ids = np.array(seqs)
    got = D.mask_labels(ids, S, E, U, A, PAD)
    assert np.array_equal(got, H.mask_labels_ref(ids, S, E, U, A, PAD))
    assert got[0].tolist()[13:17] == [11, 12, 13, E] and (got[0][:13] == -100).all()


def test_intersection_union_and_giou_ciou():
    g = gold()
    a = (torch.from_numpy(np.asarray(__import__("oracle.detweights", fromlist=["x"]).det_array("g1_iou_a", (8, 8)))) > 0).long()
    b = (torch.from_numpy(np.asarray(__import__("oracle.detweights", fromlist=["x"]).det_array("g1_iou_b", (8, 8)))) > 0.2).long()
    i, u, t = M.intersection_and_union(a.clone(), b.clone(), 2)
    assert np.array_equal(np.stack([i.numpy(), u.numpy(), t.numpy()]), g["g1_iau"])          # reference utils.utils.intersectionAndUnionGPU
    rng = torch.Generator().manual_seed(0)
    pairs = []
    for k in range(4):
        gt = (torch.rand(3, 12, 10, generator=rng) > 0.6).int()
        pr = (torch.rand(3, 12, 10, generator=rng) > 0.5).int()
        if k == 1:
            gt[0] = 0            # empty target, non-empty prediction
        if k == 2:
            gt[1] = 0; pr[1] = 0  # empty both -> IoU counts as 1
        pairs.append((pr, gt))
    acc = M.GIoUCIoU()
    for pr, gt in pairs:
        acc.update(pr, gt)
    giou, ciou = acc.compute()
    rg, rc = H.giou_ciou_ref(pairs)
    assert abs(giou - rg) < 1e-6 and abs(ciou - rc) < 1e-6


def test_db_eval_iou():
    a = np.zeros((2, 5, 5), bool); s = np.zeros((2, 5, 5), bool)
    a[0, :2] = True; s[0, 1:3] = True
    j = M.db_eval_iou(a, s)
    assert np.allclose(j, [5 / 15, 1.0])
    assert M.db_eval_iou(a[1], s[1]) == 1


def test_stom_shift_and_mean_flow():
    rng = np.random.default_rng(0)
    src = np.zeros((20, 30, 4), np.uint8)
    src[5:9, 10:16] = rng.integers(1, 255, (4, 6, 4))
    for fx, fy in [(3.7, -2.2), (-12.5, 4.0), (25.0, 0.9)]:
        assert np.array_equal(ST.shift_overlay(src, (20, 30), fx, fy), H.stom_shift_ref(src, (20, 30), fx, fy))
    vip = rng.uniform(0, 20, (40, 2))
    tgt = vip + np.array([2.0, -1.0]) + rng.normal(0, 0.05, (40, 2))
    tgt[:3] += 30  # outliers removed by the MAD filter
    vis = np.ones(40, bool); vis[5] = False
    dx, dy = ST.mean_flow(vip, tgt, vis)
    assert abs(dx - 2.0) < 0.1 and abs(dy + 1.0) < 0.1
    assert ST.mean_flow(vip, tgt, np.zeros(40, bool)) is None
    frames = [rng.integers(0, 255, (20, 30, 3), dtype=np.uint8) for _ in range(3)]
    tracks = np.stack([vip, tgt, vip])[None]
    out = ST.STOM().propagate_in_video(frames, src, 0, tracks=tracks, visibility=np.ones((1, 3, 40), bool))
    assert len(out) == 3 and out[1].shape == (20, 30, 3) and not np.array_equal(out[1], frames[1])


def test_frame_sampling_matches_reference_tables():
    """uniform_sample / get_sparse_indices / get_dense_indices (reference utils/utils.py:201-229) against tables produced by executing the
    reference's own functions (tests/golden/make_frame_sampling_fixtures.py): integer outputs, bit-exact."""
    import os

    import numpy as np

    from rga3.utils import data as D

    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frame_sampling.npz"))
    for row in G["uniform"].tolist():
        total, n = row[:2]
        assert D.uniform_sample(total, n) == row[2:2 + n], (total, n)
    for row in G["sparse"].tolist():
        total, n = row[:2]
        got = D.get_sparse_indices(total, n)
        assert got == row[2:2 + n] and got == sorted(got), (total, n)
    for row in G["dense"].tolist():
        nm, ns = row[:2]
        assert D.get_dense_indices(nm, ns) == row[2:2 + ns], (nm, ns)
    assert len(D.get_sparse_indices(3, 8)) == 8 and set(D.get_sparse_indices(3, 8)) == {0, 1, 2}
