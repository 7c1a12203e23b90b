"""CPU: the reference's import line resolves to this build when rga3-release_amd/dropin leads sys.path, and the public surface
(class / method names, config fields, output dict keys in the source) is the reference's."""
import inspect
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_line_and_surface():
    code = ("import sys; sys.path.insert(0, %r); "
            "from model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel; from model.sam2 import SAM2; from model.STOM import STOM; "
            "c = UniGRConfig(train_mask_decoder=True, out_dim=256, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seg_token_idx=7, sam_pretrained=None, hidden_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1, intermediate_size=64, vocab_size=32, vision_config=dict(depth=1, hidden_size=32, num_heads=2, intermediate_size=32, out_hidden_size=64)); "
            "assert (c.train_mask_decoder, c.out_dim, c.seg_token_idx, c.hidden_size) == (True, 256, 7, 64); "
            "m = UniGRModel(c); "
            "assert not hasattr(m, 'grounding_encoder'); "
            "names = [n for n, _ in m.named_parameters()]; "
            "assert 'model.embed_tokens.weight' in names and 'lm_head.weight' in names and any('self_attn.q_proj' in n for n in names) and any(n.startswith('visual.blocks.0.attn.qkv') for n in names); "
            "print('ok')") % os.path.join(ROOT, "rga3-release_amd", "dropin")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_method_signatures_match_reference_contract():
    from rga3.model.qwen_2_5_vl_sam2 import UniGRModel

    ev = list(inspect.signature(UniGRModel.evaluate).parameters)
    assert ev == ["self", "input_ids", "attention_mask", "pixel_values", "pixel_values_videos", "image_grid_thw", "video_grid_thw",
                  "second_per_grid_ts", "images_sam", "resize_list", "original_size_list"]
    mf = inspect.signature(UniGRModel.model_forward).parameters
    for k in ("input_ids", "attention_mask", "position_ids", "labels", "pixel_values_videos", "video_grid_thw", "second_per_grid_ts", "images_sam",
              "offset", "masks_list", "label_list", "resize_list", "inference", "kwargs"):
        assert k in mf, k
    src = inspect.getsource(UniGRModel.model_forward)
    for key in ('"loss"', '"ce_loss"', '"mask_bce_loss"', '"mask_dice_loss"', '"mask_loss"', '"pred_masks"', '"gt_masks"'):
        assert key in src
    assert list(inspect.signature(UniGRModel.forward).parameters) == ["self", "kwargs"]


def test_gradient_checkpointing_flag_drives_the_recompute_switch():
    """reference train_joint.py:188 calls model.gradient_checkpointing_enable(): here that IS the decoder's activation-recompute switch."""
    from rga3.model import qwen_train as QT
    from rga3.model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel

    c = UniGRConfig(train_mask_decoder=True, out_dim=256, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seg_token_idx=7, sam_pretrained=None,
                    hidden_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1, intermediate_size=64, vocab_size=32,
                    vision_config=dict(depth=1, hidden_size=32, num_heads=2, intermediate_size=32, out_hidden_size=64))
    m = UniGRModel(c)
    assert QT._acts["store"] is True and not m.is_gradient_checkpointing
    try:
        m.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"use_reentrant": False})
        assert QT._acts["store"] is False and m.is_gradient_checkpointing
        m.gradient_checkpointing_disable()
        assert QT._acts["store"] is True and not m.is_gradient_checkpointing
    finally:
        QT.set_activation_recompute(False)


def test_heads_output_keeps_the_reference_keys_with_ious_made_on_first_read():
    """forward_sam_heads' inference result (reference model/sam2.py:3262-3431 returns low_res_multimasks, ious, low_res_masks, high_res_masks, obj_ptr,
    object_score_logits): "ious" is converted from the decoder's bf16 IoU logits when it is first read -- same key, same value, every dict access form."""
    import torch

    from rga3.model.sam2 import _HeadsOut

    iou = torch.arange(8, dtype=torch.float32).reshape(2, 4).to(torch.bfloat16)
    o = _HeadsOut({"low_res_masks": 1, "obj_ptr": 2}, iou)
    assert "ious" in o and "low_res_masks" in o and "nope" not in o
    assert o.get("nope") is None and o.get("obj_ptr") == 2
    want = iou[:, 1:].float()
    assert o["ious"].dtype == torch.float32 and torch.equal(o["ious"], want) and torch.equal(o.get("ious"), want)
    assert o["ious"] is o["ious"]                       # made once
    try:
        o["missing"]
    except KeyError:
        pass
    else:
        raise AssertionError("unknown keys must raise KeyError")
