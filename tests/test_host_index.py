"""CPU: the product's host-side integer plumbing (rga3/model/qwen_index.py) is bit-exact against the golden
vectors from transformers 5.15 and against the oracle's independent loop implementation."""
import os

import numpy as np
import pytest

from oracle import qwen25vl as Q
from rga3.model import qwen_index as QI

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen_tiny.npz"), allow_pickle=False)


@pytest.mark.parametrize("key", ["a", "b", "c", "full16"])
def test_vision_plan(key):
    g = GOLD[f"g4_{key}_grid"]
    wi, cw = QI.vision_window_index(g, 2, 112, 14)
    assert wi.dtype == np.int64 and cw.dtype == np.int32
    assert np.array_equal(wi, GOLD[f"g4_{key}_window_index"])
    assert np.array_equal(cw, GOLD[f"g4_{key}_cu_window"])
    assert np.array_equal(QI.vision_cu_seqlens(g), GOLD[f"g4_{key}_cu_full"])
    assert np.array_equal(QI.vision_position_ids(g, 2), GOLD[f"g4_{key}_pos_ids"])


@pytest.mark.parametrize("key", ["a", "b"])
@pytest.mark.parametrize("rule", ["hf515", "hf449"])
def test_rope_index(key, rule):
    pos, d = QI.rope_index(GOLD[f"rope_{key}_input_ids"], 301, 302, 2, 2, None, GOLD[f"rope_{key}_grid"], GOLD[f"rope_{key}_spg"],
                           GOLD[f"rope_{key}_attention_mask"], rule)
    assert np.array_equal(pos, GOLD[f"rope_{key}_position_ids"])
    assert np.array_equal(d, GOLD[f"rope_{key}_deltas"])


def test_random_grids_match_oracle():
    rng = np.random.default_rng(0)
    for _ in range(25):
        n = rng.integers(1, 3)
        g = np.stack([rng.integers(1, 4, n), 2 * rng.integers(1, 14, n), 2 * rng.integers(1, 14, n)], 1)
        a, b = QI.vision_window_index(g, 2, 112, 14), Q.vision_window_index(g, 2, 112, 14)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(QI.vision_position_ids(g, 2), Q.vision_position_ids(g, 2))
        assert np.array_equal(QI.vision_cu_seqlens(g), Q.vision_cu_seqlens(g))


def test_rope_index_text_only_and_images():
    cfg = Q.QwenCfg()
    cfg.image_token_id, cfg.video_token_id = 301, 302
    ids = np.array([[5, 6, 301, 301, 301, 301, 7, 8, 302, 302, 302, 302, 302, 302, 302, 302, 9]])
    kw = dict(image_grid_thw=[[1, 4, 4]], video_grid_thw=[[2, 4, 4]], second_per_grid_ts=[0.5])
    for rule in ("hf449", "hf515"):
        a = QI.rope_index(ids, 301, 302, 2, 2, attention_mask=None, temporal_rule=rule, **kw)
        b = Q.rope_index(ids, cfg, attention_mask=None, temporal_rule=rule, **kw)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    t = QI.rope_index(np.array([[1, 2, 3, 4]]), 301, 302, 2, 2)
    assert np.array_equal(t[0][:, 0], np.tile(np.arange(4), (3, 1)))
