"""CPU: pins the Qwen2.5-VL oracle (oracle/qwen25vl.py) against golden vectors produced by the installed
transformers 5.15.0 implementation (tests/golden/make_qwen_fixtures.py).  Integer outputs bit-exact; fp32 <= 1e-4."""
import os

import numpy as np
import pytest
import torch

from oracle import qwen25vl as Q
from oracle.detweights import det_state_dict, det_tensor

GOLD = os.path.join(os.path.dirname(__file__), "golden", "qwen_tiny.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


def tiny_cfg():
    return Q.QwenCfg(
        vision=Q.VisionCfg(depth=4, hidden_size=64, num_heads=4, intermediate_size=88, patch_size=14, temporal_patch_size=2,
                           spatial_merge_size=2, window_size=112, fullatt_block_indexes=(1, 3), out_hidden_size=96,
                           in_channels=3, tokens_per_second=2),
        text=Q.TextCfg(hidden_size=96, num_hidden_layers=2, num_attention_heads=6, num_key_value_heads=2, intermediate_size=160,
                       vocab_size=320, rms_norm_eps=1e-6, rope_theta=1000000.0, mrope_section=(2, 3, 3)),
        image_token_id=301, video_token_id=302, vision_start_token_id=303)


def tiny_params(gold):
    shapes = {str(n): eval(str(s)) for n, s in zip(gold["param_names"], gold["param_shapes"])}
    return det_state_dict(shapes, seed=1)


@pytest.mark.parametrize("key", ["a", "b", "c", "full16"])
def test_vision_index_plumbing_bit_exact(gold, key):
    g = gold[f"g4_{key}_grid"]
    wi, cw = Q.vision_window_index(g, 2, 112, 14)
    assert np.array_equal(wi, gold[f"g4_{key}_window_index"])
    assert np.array_equal(cw, gold[f"g4_{key}_cu_window"])
    assert np.array_equal(Q.vision_cu_seqlens(g), gold[f"g4_{key}_cu_full"])
    assert np.array_equal(Q.vision_position_ids(g, 2), gold[f"g4_{key}_pos_ids"])


@pytest.mark.parametrize("key", ["a", "b"])
@pytest.mark.parametrize("rule", ["hf515", "hf449"])
def test_rope_index_bit_exact(gold, key, rule):
    pos, delta = Q.rope_index(gold[f"rope_{key}_input_ids"], tiny_cfg(), None, gold[f"rope_{key}_grid"], gold[f"rope_{key}_spg"],
                              gold[f"rope_{key}_attention_mask"], temporal_rule=rule)
    assert np.array_equal(pos, gold[f"rope_{key}_position_ids"])
    assert np.array_equal(delta, gold[f"rope_{key}_deltas"])


@pytest.mark.parametrize("key", ["a", "b"])
def test_vit_forward(gold, key):
    cfg, P = tiny_cfg(), tiny_params(gold)
    g = gold[f"g4_{key}_grid"]
    px = det_tensor(f"pixel_values_{key}", (int(np.prod(g[0])), 1176), 1.0, seed=5)
    out, pre = Q.vit_forward(P, px, g, cfg, return_pre_merge=True)
    assert np.abs(out.numpy() - gold[f"vit_{key}_pooler"]).max() < 1e-4
    assert np.abs(pre.numpy() - gold[f"vit_{key}_last_hidden"]).max() < 1e-4


def test_full_forward_and_loss(gold):
    cfg, P = tiny_cfg(), tiny_params(gold)
    px = torch.cat([det_tensor("pixel_values_full0", (192, 1176), 1.0, seed=5), det_tensor("pixel_values_full1", (192, 1176), 1.0, seed=6)], 0)
    r = Q.forward(P, cfg, torch.from_numpy(gold["full_input_ids"]), torch.from_numpy(gold["full_attention_mask"]),
                  labels=torch.from_numpy(gold["full_labels"]), pixel_values_videos=px, video_grid_thw=gold["full_grid"],
                  second_per_grid_ts=np.array([1.0, 1.0]))
    assert np.array_equal(r["position_ids"].numpy(), gold["full_position_ids"])
    am = gold["full_attention_mask"].astype(bool)
    assert np.abs(r["hidden"].numpy() - gold["full_hidden_last"])[am].max() < 2e-4
    assert np.abs(r["logits"].numpy() - gold["full_logits"])[am].max() < 2e-4
    assert abs(float(r["loss"]) - float(gold["full_loss"])) < 1e-5
