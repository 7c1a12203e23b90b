"""CPU: the restated Pillow bicubic resample (oracle/preproc.py) is bit-exact against Pillow itself and against the committed
fixture made from the reference's DirectResize / preprocess recipe (tests/golden/make_preproc_fixtures.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import preproc as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("hw", [(480, 854), (720, 1280), (360, 640), (1080, 1920), (256, 256), (1500, 700), (1024, 1024), (37, 2000), (1024, 999)])
def test_resize_matches_pillow(hw):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    img = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    img[: hw[0] // 3, :, 0] = 255   # saturated regions exercise the clip
    img[hw[0] // 2:, : hw[1] // 4, :] = 0
    want = np.array(Image.fromarray(img, "RGB").resize((1024, 1024)))
    got = P.resize_bicubic_u8(img, 1024, 1024)
    assert got.dtype == np.uint8 and got.shape == (1024, 1024, 3)
    assert np.array_equal(got, want), int(np.abs(got.astype(int) - want.astype(int)).max())


def test_small_targets_and_coeff_table():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(90, 130, 3), dtype=np.uint8)
    for size in [(64, 64), (200, 50), (90, 260)]:
        want = np.array(Image.fromarray(img, "RGB").resize((size[1], size[0])))
        assert np.array_equal(P.resize_bicubic_u8(img, size[0], size[1]), want), size
    b, kk, ksize = P.pil_bicubic_coeffs(1920, 1024)
    assert ksize == 9 and kk.shape == (1024, 9) and int(b[:, 1].max()) <= 9
    assert np.all(np.abs(kk.sum(1) - (1 << P.PRECISION_BITS)) <= ksize)   # rows sum to 1.0 in fixed point up to rounding


def test_fixture_from_reference_recipe():
    G = np.load(os.path.join(ROOT, "tests", "golden", "preproc.npz"))
    for i in range(int(G["n"])):
        img = G[f"img{i}"]
        res, norm = P.sam_preprocess(img[None], size=int(G["L"]))
        assert np.array_equal(res[0], G[f"res{i}"])
        got = norm[0].to(torch.bfloat16).float().numpy()
        assert np.array_equal(got[:, ::4, ::4], G[f"norm_bf16_sub{i}"])


# ---------------------------------------------------------------------------------------------- Qwen side
def _qg():
    return np.load(os.path.join(ROOT, "tests", "golden", "qwen_preproc.npz"))


def test_smart_resize_matches_transformers_table():
    """400 (h, w, min, max) -> (h', w') rows produced by the installed transformers' smart_resize (make_qwen_preproc_fixtures.py)."""
    tab = _qg()["smart_resize"]
    for h, w, mn, mx, oh, ow in tab.tolist():
        assert P.smart_resize(h, w, 28, mn, mx) == (oh, ow), (h, w, mn, mx)
    with pytest.raises(ValueError):
        P.smart_resize(10, 2100)


def test_qwen_video_recipe_matches_fixture():
    """Pillow resize + HF patchify + both normalisation orders, incl. an odd frame count and the second (processor-side) resize."""
    G = _qg()
    for i in range(int(G["n"])):
        for fused, key in ((False, "pv49_"), (True, "pv5_")):
            pv, grid, res = P.qwen_video_preprocess(G[f"frames{i}"], max_pixels=int(G[f"max_pixels{i}"]), fused=fused)
            assert np.array_equal(res, G[f"res{i}"]), i
            assert list(grid) == G[f"grid{i}"].tolist()
            assert pv.dtype == np.float32 and np.array_equal(pv, G[f"{key}{i}"]), (i, fused)


def test_normalisation_orders_agree_in_bf16():
    """The reference casts pixel_values_videos to bf16 (inference_mevis.py:214, train_joint.py:516-519): the 4.49 and 5.x orders differ by
    a few fp32 ulps before the cast (<= 2.4e-7 absolute) and on NONE of the 768 (channel, byte) values after it."""
    a, b = torch.from_numpy(P.qwen_norm_lut(fused=False)), torch.from_numpy(P.qwen_norm_lut(fused=True))
    assert float((a - b).abs().max()) <= 5e-7 and not torch.equal(a, b)
    assert torch.equal(a.bfloat16(), b.bfloat16())
