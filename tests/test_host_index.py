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


def test_tuner_refine_keeps_what_makes_the_step_fastest(monkeypatch):
    """Host logic of the in-situ tile refinement (rga3/hip/tuner.py::refine) with a fake clock: a candidate that is slower stand-alone but
    makes the step faster is adopted; candidates far off the stand-alone best are not tried; a gain under the threshold changes nothing."""
    from rga3.hip import tuner

    monkeypatch.setattr(tuner, "_enabled", True)
    monkeypatch.setattr(tuner, "_cache", {"a": 3, "b": 21, "c": 4})
    monkeypatch.setattr(tuner, "_times", {"a": {3: 1.0, 11: 1.1, 22: 9.0}, "b": {21: 2.0, 20: 2.05}, "c": {4: 1.0, 12: 1.2}})
    cost = {("a", 3): 5.0, ("a", 11): 4.0, ("a", 22): 1.0, ("b", 21): 7.0, ("b", 20): 7.2, ("c", 4): 3.0, ("c", 12): 2.999}
    tried = []

    def fake_time(step, reps):
        tried.append(dict(tuner._cache))
        return sum(cost[(k, t)] for k, t in tuner._cache.items())

    monkeypatch.setattr(tuner, "_time_step", fake_time)
    monkeypatch.setattr(tuner.torch.cuda, "synchronize", lambda: None)
    ch = tuner.refine(lambda: None, reps=1, within=1.5, min_gain=0.002)
    assert ch == {"a": (3, 11)} and tuner._cache == {"a": 11, "b": 21, "c": 4}
    assert all(t["a"] != 22 for t in tried)          # 9x the stand-alone best: never tried, although it would have won
