"""CPU: checkpoint interop -- HF directory layout round trip (single file and sharded index), PEFT-wrapped key
names, the strict / late-module rules of from_pretrained, and the LoRA merge formula.  (No GPU: parameters are only copied.)"""
import json
import os

import pytest
import torch

TINY = dict(train_mask_decoder=True, out_dim=256, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seg_token_idx=7, sam_pretrained=None,
            hidden_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, intermediate_size=64, vocab_size=32,
            vision_config=dict(depth=1, hidden_size=32, num_heads=2, intermediate_size=32, out_hidden_size=64))


def _model(seed=0):
    from rga3.model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel

    torch.manual_seed(seed)
    m = UniGRModel(UniGRConfig(**TINY))
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.1)
    return m


@pytest.mark.parametrize("shard_bytes", [1 << 40, 20_000])
def test_save_load_round_trip(tmp_path, shard_bytes):
    from rga3.model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel

    m = _model(1).to(torch.bfloat16)
    m.save_pretrained(str(tmp_path), max_shard_size=shard_bytes)
    files = sorted(os.listdir(tmp_path))
    assert "config.json" in files
    if shard_bytes < 1 << 30:
        idx = json.load(open(tmp_path / "model.safetensors.index.json"))
        assert len(set(idx["weight_map"].values())) > 1 and set(idx["weight_map"]) == set(m.state_dict())
    else:
        assert "model.safetensors" in files
    cfg = UniGRConfig.from_pretrained(str(tmp_path), train_mask_decoder=True)
    m2 = UniGRModel.from_pretrained(str(tmp_path), config=cfg, torch_dtype=torch.bfloat16, attn_implementation="flash_attention_2", low_cpu_mem_usage=False)
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert set(sd) == set(sd2)
    for k in sd:
        assert sd2[k].dtype == torch.bfloat16 and torch.equal(sd[k], sd2[k]), k


def test_strictness_and_peft_prefixes(tmp_path):
    from safetensors.torch import save_file

    from rga3.model.qwen_2_5_vl_sam2 import UniGRModel
    from rga3.utils import checkpoint as CK

    m = _model(2)
    sd = {("base_model.model." + k).replace("q_proj.weight", "q_proj.base_layer.weight"): v.clone() for k, v in m.state_dict().items()}
    save_file(sd, str(tmp_path / "model.safetensors"))
    m2 = _model(3)
    missing, unexpected = CK.load_checkpoint(m2, str(tmp_path))
    assert not missing and not unexpected
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    # an unknown key or a missing base weight is an error; modules created after from_pretrained (SAM2, text_hidden_fcs, LoRA) are not
    bad = dict(m.state_dict())
    bad.pop("lm_head.weight")
    save_file({k: v.clone() for k, v in bad.items()}, str(tmp_path / "model.safetensors"))
    with pytest.raises(RuntimeError):
        CK.load_checkpoint(_model(4), str(tmp_path))
    m5 = _model(5)
    m5.initialize_sam_modules(m5.config) if False else None
    bad2 = dict(m.state_dict())
    bad2["not.a.parameter"] = torch.zeros(1)
    save_file({k: v.clone() for k, v in bad2.items()}, str(tmp_path / "model.safetensors"))
    with pytest.raises(RuntimeError):
        CK.load_checkpoint(_model(6), str(tmp_path))
    assert isinstance(UniGRModel.from_pretrained(str(tmp_path), config=m.config, strict=False), UniGRModel)


def test_lora_merge_formula_and_names():
    from rga3.model.qwen_train import LoRALinear, add_lora
    from rga3.utils import checkpoint as CK

    m = _model(7)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    hits = add_lora(m, r=4, alpha=8)
    assert hits and all(("q_proj" in h or "v_proj" in h) for h in hits)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lora_" in n:
                p.normal_(0, 0.2)
    lora = {n: p.detach().clone() for n, p in m.named_parameters() if "lora_" in n}
    merged = m.merge_and_unload()
    assert merged is m and not any(isinstance(x, LoRALinear) for x in m.modules())
    after = m.state_dict()
    assert set(after) == set(before)                      # PEFT's merge_and_unload leaves the base model's names
    for h in hits:
        A, B = lora[h + ".lora_A.default.weight"], lora[h + ".lora_B.default.weight"]
        want = before[h + ".weight"].float() + (8 / 4) * (B.float() @ A.float())
        assert torch.allclose(after[h + ".weight"].float(), want, atol=1e-6), h
    for k in before:
        if not any(k == h + ".weight" for h in hits):
            assert torch.equal(before[k], after[k]), k


def test_tied_word_embeddings_round_trip(tmp_path):
    """Qwen2.5-VL-3B ties lm_head to embed_tokens and HF omits the tied tensor from the checkpoint: strict load must accept that, the
    module must alias the two, and save must store the tensor once."""
    from safetensors import safe_open

    from rga3.model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel

    cfg = dict(TINY, tie_word_embeddings=True)
    torch.manual_seed(4)
    m = UniGRModel(UniGRConfig(**cfg))
    assert m.lm_head.weight is m.model.embed_tokens.weight
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.1)
    m.save_pretrained(str(tmp_path))
    with safe_open(str(tmp_path / "model.safetensors"), "pt") as f:
        keys = set(f.keys())
    assert "model.embed_tokens.weight" in keys and "lm_head.weight" not in keys
    m2 = UniGRModel.from_pretrained(str(tmp_path), config=UniGRConfig.from_pretrained(str(tmp_path), train_mask_decoder=True))
    assert m2.lm_head.weight is m2.model.embed_tokens.weight
    assert torch.equal(m2.lm_head.weight, m.model.embed_tokens.weight)
    m2.resize_token_embeddings(40)
    assert m2.lm_head.weight is m2.model.embed_tokens.weight and m2.lm_head.weight.shape[0] == 40


def test_load_sam2_checkpoint_pt(tmp_path):
    """The SAM2 .pt loader of reference model/sam2.py:30-85: {'model': sd} / {'state_dict': sd} / bare sd, '.gamma' -> '.g_weight' (and only with the
    leading dot), strict on missing and unexpected keys."""
    from rga3.model.sam2 import SAM2, load_sam2_checkpoint

    tiny = dict(image_size=128, embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4), memattn_layers=2)
    torch.manual_seed(5)
    src = SAM2(**tiny).sam2_model
    with torch.no_grad():
        for p in src.parameters():
            p.normal_(0, 0.1)
    sd = src.state_dict()
    gkeys = [k for k in sd if k.endswith(".g_weight")]
    assert gkeys, "the memory encoder's layer-scale parameters are stored as .g_weight"
    upstream = {k.replace(".g_weight", ".gamma"): v.clone() for k, v in sd.items()}   # what sam2_hiera_*.pt holds
    for wrap in ("model", "state_dict", None):
        path = str(tmp_path / f"sam2_{wrap}.pt")
        torch.save({wrap: upstream} if wrap else upstream, path)
        dst = SAM2(**tiny).sam2_model
        load_sam2_checkpoint(dst, path)
        for k, v in sd.items():
            assert torch.equal(dst.state_dict()[k], v), k
    # through the wrapper's constructor argument, as initialize_sam_modules passes config.sam_pretrained (reference qwen_2_5_vl_sam2.py:119)
    m = SAM2(ckpt_path=str(tmp_path / "sam2_model.pt"), **tiny)
    assert torch.equal(m.sam2_model.state_dict()[gkeys[0]], sd[gkeys[0]])
    # strictness
    bad = dict(upstream)
    bad.pop(next(iter(bad)))
    torch.save({"model": bad}, str(tmp_path / "missing.pt"))
    with pytest.raises(RuntimeError):
        load_sam2_checkpoint(SAM2(**tiny).sam2_model, str(tmp_path / "missing.pt"))
    extra = dict(upstream, **{"not.a.key": torch.zeros(1)})
    torch.save({"model": extra}, str(tmp_path / "extra.pt"))
    with pytest.raises(RuntimeError):
        load_sam2_checkpoint(SAM2(**tiny).sam2_model, str(tmp_path / "extra.pt"))
