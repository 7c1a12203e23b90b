"""CPU: the host-only coefficient entry point of the C ABI reproduces the oracle's (= Pillow's) fixed-point tables."""
import numpy as np
import pytest

from oracle import preproc as P


@pytest.mark.parametrize("sizes", [(854, 1024), (480, 1024), (1920, 1024), (1080, 1024), (1024, 256), (37, 1024), (2000, 1024), (999, 1024)])
def test_coeff_tables_match_oracle(sizes):
    from rga3.utils.preproc import pil_bicubic_tables

    b, k = pil_bicubic_tables(*sizes)
    ob, ok_, ks = P.pil_bicubic_coeffs(*sizes)
    assert k.shape[1] == ks
    assert np.array_equal(b.numpy(), ob)
    assert np.array_equal(k.numpy(), ok_)


def test_host_smart_resize_and_lut_match_oracle():
    """Host logic of the Qwen-side pipeline: smart_resize over the transformers-made table; the C ABI's byte table in both orders."""
    import os

    import torch

    from rga3.utils import preproc as H

    tab = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "qwen_preproc.npz"))["smart_resize"]
    for h, w, mn, mx, oh, ow in tab.tolist():
        assert H.smart_resize(h, w, 28, mn, mx) == (oh, ow)
    for fused in (False, True):
        lut = H.qwen_norm_lut(torch.device("cpu"), fused=fused)
        assert np.array_equal(lut.numpy(), P.qwen_norm_lut(fused=fused)), fused
