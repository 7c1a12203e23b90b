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


def test_stom_mask_shape_branch():
    """reference STOM.py:163-207 restated without cv2: the structuring elements equal OpenCV's documented 3x3 / 5x5 ellipses; closing bridges gaps
    narrower than the element and leaves isolated far points apart; the filled circle is the midpoint raster (extent exactly r on both axes,
    symmetric); warp_point draws it at the centroid of the closed point cloud in the prompt's colour with alpha clamped to [96, 148]."""
    assert ST.ellipse_kernel(3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert ST.ellipse_kernel(5).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    rng0 = np.random.default_rng(3)
    for k in (5, 6):                       # odd and even elements (even ones are not symmetric about the anchor: OpenCV applies them as they are)
        m = np.zeros((24, 31), np.uint8)
        m[rng0.integers(0, 24, 40), rng0.integers(0, 31, 40)] = 255
        m[10:14, 12] = 255
        ker = ST.ellipse_kernel(k)
        a = k // 2
        offs = [(i - a, j - a) for i in range(k) for j in range(k) if ker[i, j]]
        inside = lambda y, x: 0 <= y < 24 and 0 <= x < 31
        dil = np.zeros_like(m)
        for y in range(24):
            for x in range(31):
                dil[y, x] = max([m[y + dy, x + dx] for dy, dx in offs if inside(y + dy, x + dx)] or [0])
        clo = np.zeros_like(m)
        for y in range(24):
            for x in range(31):
                clo[y, x] = min([dil[y + dy, x + dx] for dy, dx in offs if inside(y + dy, x + dx)] or [255])
        assert np.array_equal(ST.morph_close(m, k), clo), k
        if k % 2:
            assert (ST.morph_close(m, k) >= m).all()       # closing is extensive for an element symmetric about its anchor
    circ = ST.filled_circle((50, 50), 25, 20, 7)
    assert circ[20, 18:33].min() == 255 and circ[20, 17] == 0 and circ[20, 33] == 0 and circ[13, 25] == 255 and circ[12, 25] == 0 and circ[27, 25] == 255 and circ[28, 25] == 0
    assert np.array_equal(circ[:, 18:33], circ[:, 18:33][:, ::-1])
    assert np.array_equal(circ[13:28], circ[13:28][::-1])
    clipped = ST.filled_circle((30, 30), 2, 3, 6)            # partly outside: same raster, cut at the border
    full = ST.filled_circle((60, 60), 32, 33, 6)
    assert np.array_equal(clipped, full[30:60, 30:60])
    # warp_point: 40 points around (row 30, col 45) in a 60 x 90 frame
    rng = np.random.default_rng(0)
    h, w = 60, 90
    src = np.zeros((h, w, 4), np.uint8)
    src[10:20, 10:20] = (200, 30, 40, 255)
    tgt = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
    pts = np.stack([45 + rng.integers(-2, 3, 40), 30 + rng.integers(-2, 3, 40)], 1).astype(np.float32)   # (col, row) as the tracker returns them
    vis = np.ones(40, bool)
    out = ST.warp_point(src, tgt, pts, vis)
    k, r = min(h, w) // 15, min(h, w) // 20
    closed = ST.morph_close(np.where(np.isin(np.arange(h * w).reshape(h, w), (pts[:, 1].astype(int) * w + pts[:, 0].astype(int))), 255, 0).astype(np.uint8), k)
    ys, xs = np.nonzero(closed)
    cx, cy = int(xs.mean()), int(ys.mean())
    disk = ST.filled_circle((h, w), cx, cy, r) > 0
    assert np.array_equal(out[~disk], tgt[~disk])
    a = 148 / 255.0
    exp = np.array(Image_alpha(tgt, disk, (200, 30, 40, 148)))
    assert np.array_equal(out, exp)
    assert np.array_equal(ST.warp_point(src, tgt, pts, np.arange(40) < 10), tgt)      # fewer than half visible: frame untouched
    frames = [tgt, tgt.copy(), tgt.copy()]
    tracks = np.broadcast_to(pts[None, None], (1, 3, 40, 2)).copy()
    res = ST.STOM().propagate_in_video(frames, src, 0, shape="mask", tracks=tracks, visibility=np.ones((1, 3, 40), bool))
    assert len(res) == 3 and np.array_equal(res[1], out) and np.array_equal(res[2], out)


def Image_alpha(tgt, disk, rgba):
    from PIL import Image
    ov = np.zeros(tgt.shape[:2] + (4,), np.uint8)
    ov[disk] = rgba
    return Image.alpha_composite(Image.fromarray(tgt, "RGB").convert("RGBA"), Image.fromarray(ov, "RGBA")).convert("RGB")


def test_bench_refuses_counter_profiles_of_another_tree(tmp_path, monkeypatch):
    """bench.py quotes a PMC traffic / pipe-utilisation profile only when it was collected on the running tree (fingerprint of the HIP sources + the
    package's Python, stamped by tools/pmc_traffic.py / pmc_pipe_util.py); anything else is reported as `traffic: null, traffic_stale: true`, never as a number."""
    import importlib
    import json

    bench = importlib.import_module("bench")
    from rga3.utils.fingerprint import tree_fingerprint

    fp = tree_fingerprint()
    assert len(fp) == 16 and fp == tree_fingerprint()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    roof = {}
    bench._put_traffic(roof, "forward")
    assert roof == {"traffic": None}                                     # nothing collected: null, not stale
    (prof / f"{bench.PROFILE_ROUND}_bench_forward_gemm_traffic.json").write_text(json.dumps({"traffic_bytes_per_launch": 123456789.4, "tree": "0123456789abcdef"}))
    roof = {}
    bench._put_traffic(roof, "forward")
    assert roof["traffic"] is None and roof["traffic_stale"] is True and "REFUSED" in roof["traffic_source"]
    (prof / f"{bench.PROFILE_ROUND}_bench_forward_gemm_traffic.json").write_text(json.dumps({"traffic_bytes_per_launch": 123456789.4, "tree": fp}))
    roof = {}
    bench._put_traffic(roof, "forward")
    assert roof["traffic"] == 123456789 and "traffic_stale" not in roof and fp in roof["traffic_source"]
    # a counter pass that recorded no launches (the profiler died) is no figure either -- never a zero
    (prof / f"{bench.PROFILE_ROUND}_bench_forward_gemm_traffic.json").write_text(json.dumps({"traffic_bytes_per_launch": 0.0, "launches_fetch_pass": 0, "launches_write_pass": 0, "tree": fp}))
    roof = {}
    bench._put_traffic(roof, "forward")
    assert roof["traffic"] is None and "no launches" in roof["traffic_source"]
    (prof / f"{bench.PROFILE_ROUND}_pmc_pipe_util.json").write_text(json.dumps({"gemm_nt_sk_kernel<2,": {"mfma_busy": 0.56}, "_tree": "feedfeedfeedfeed"}))
    mb, src = bench._mfma_busy("gemm_nt")
    assert mb is None and "REFUSED" in src
    (prof / f"{bench.PROFILE_ROUND}_pmc_pipe_util.json").write_text(json.dumps({"gemm_nt_sk_kernel<2,": {"mfma_busy": 0.56}, "_tree": fp}))
    mb, src = bench._mfma_busy("gemm_nt")
    assert mb == {"gemm_nt_sk_kernel<2,": 0.56}


def test_oracle_gate_up_column_order_is_the_builds_interleave():
    """oracle/fp8step.py::fp8_frozen_linears_on_codes re-orders the columns of [Wg; Wu]^T to the build's gate | up pack (16-row blocks alternating, the layout of
    rga3_gemm_swiglu_pre_bf16's weight): its index map must be exactly rga3.model.qwen2_5_vl._interleave_rows applied to the row numbers."""
    import torch

    from rga3.model.qwen2_5_vl import _interleave_rows

    for I in (16, 64, 18944):
        rows_g, rows_u = torch.arange(I).view(I, 1), (torch.arange(I) + I).view(I, 1)
        packed = _interleave_rows(rows_g, rows_u, I).view(-1)
        blk = torch.arange(I).view(I // 16, 16)
        order_b = torch.stack([blk, blk + I], 1).reshape(-1)
        assert torch.equal(packed, order_b)


def test_board_sample_never_spawns_under_a_profiler(monkeypatch, tmp_path):
    """bench.py reads board power / shader clock from sysfs hwmon in-process; rocm-smi (a '#!/usr/bin/env python3' script: an exec hop with the GPU
    already initialised by a profiler's preloaded library) is a fallback only, never under a profiler's environment."""
    import importlib
    import os
    import subprocess

    bench = importlib.import_module("bench")
    for var in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_ATT_LIBRARY_PATH"):
        for k in list(os.environ):
            if k == "LD_PRELOAD" or k.startswith(("ROCP", "HSA_TOOLS")):
                monkeypatch.delenv(k, raising=False)
        assert not bench._under_profiler()
        monkeypatch.setenv(var, "x")
        assert bench._under_profiler(), var
        monkeypatch.delenv(var)
    # no sysfs cards + profiler environment: no sample and no child process
    monkeypatch.setattr(bench, "_sysfs_board", lambda: [])
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    called = []
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: called.append(a) or (_ for _ in ()).throw(AssertionError("spawned")))
    assert bench._board_sample(lambda: None, seconds=0.1) is None and not called
    # sysfs present: read in-process (the busiest card), still no child
    pw, fq = tmp_path / "power1_average", tmp_path / "freq1_input"
    pw.write_text("1310000000\n")
    fq.write_text("2100000000\n")
    monkeypatch.setattr(bench, "_sysfs_board", lambda: [(str(pw), str(fq))])
    monkeypatch.setattr(bench.torch.cuda, "synchronize", lambda *a, **k: None)
    got = bench._board_sample(lambda: None, seconds=1.5)
    assert got is not None and got["source"] == "sysfs hwmon" and got["power_w"][0] == 1310.0 and got["sclk_mhz"][0] == 2100 and not called


def test_stom_closing_against_an_independent_library():
    """VERDICT r5 weak 4 (cv2 is not in the image: the OpenCV restatement of reference STOM.py:186-200 is unpinned against cv2 itself).  Second opinion from SciPy's
    grey morphology -- an independent implementation of the same definition (max / min over the element's support, outside pixels never win) -- for the odd,
    anchor-symmetric ellipses the reference's `min(h, w) // 15` produces on typical frames, and SciPy's centre of mass for the centroid the circle is drawn at."""
    import numpy as np
    from scipy import ndimage

    rng = np.random.default_rng(11)
    for k, shape in ((3, (45, 60)), (5, (75, 100)), (7, (105, 140)), (9, (135, 180)), (15, (240, 225))):
        m = np.zeros(shape, np.uint8)
        m[rng.integers(0, shape[0], 90), rng.integers(0, shape[1], 90)] = 255
        m[shape[0] // 3: shape[0] // 3 + k, shape[1] // 2] = 255
        m[0, :7] = 255
        m[-1, -5:] = 255                                  # border contact: outside pixels must not win the erosion
        ker = ST.ellipse_kernel(k).astype(bool)
        dil = ndimage.grey_dilation(m, footprint=ker, mode="constant", cval=0)
        want = ndimage.grey_erosion(dil, footprint=ker, mode="constant", cval=255)
        got = ST.morph_close(m, k)
        assert np.array_equal(got, want), k
        ys, xs = np.nonzero(got)
        cy, cx = ndimage.center_of_mass(got)
        assert abs(cy - ys.mean()) < 1e-9 and abs(cx - xs.mean()) < 1e-9      # cv2.moments m01 / m00, m10 / m00 of a 0 / 255 image = the mean coordinates
