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
