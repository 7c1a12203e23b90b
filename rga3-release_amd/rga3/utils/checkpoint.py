"""Checkpoint interop: read / write the Hugging Face on-disk layout the reference uses

  * `UniGRModel.from_pretrained(path, config=..., torch_dtype=...)`            reference train_joint.py:171-184, evaluation/*/inference_*.py
  * sharded `model.safetensors.index.json` / `pytorch_model.bin.index.json`     merge_lora_weights_and_save_hf_model.py:124-131
  * `model.merge_and_unload()` + `save_pretrained(path, state_dict=...)`        merge_lora_weights_and_save_hf_model.py:133-134 (PEFT)
  * SAM2 `.pt` with the `.gamma -> .g_weight` rename                              model/sam2.py:60-85 (rga3.model.sam2.load_sam2_checkpoint)

Parameter names of this build equal the reference's (strict load is a test), so loading is name-for-name; PEFT's
`base_model.model.` prefix and `.base_layer.` infix are accepted.  PEFT and DeepSpeed are not in the image: the merge follows PEFT's
published formula W += (alpha / r) * B @ A and is parity-unpinned; a DeepSpeed ZeRO-2 checkpoint DIRECTORY (what the reference trainer saves,
train_joint.py:426-461) is consolidated by rga3.utils.zero_ckpt (this build's own DDP optimizer keeps whole parameters on every rank and never writes one)."""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Optional

import torch

SAFE_INDEX = "model.safetensors.index.json"
BIN_INDEX = "pytorch_model.bin.index.json"


def _load_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu", weights_only=True)


def iter_checkpoint_shards(path: str) -> Iterable[Dict[str, torch.Tensor]]:
    """Yield state-dict shards of an HF checkpoint directory (or a single weights file), one at a time (a 7B model is 16 GB in bf16)."""
    if os.path.isfile(path):
        yield _load_file(path)
        return
    for index in (SAFE_INDEX, BIN_INDEX):
        f = os.path.join(path, index)
        if os.path.exists(f):
            with open(f) as fh:
                files = sorted(set(json.load(fh)["weight_map"].values()))
            for name in files:
                yield _load_file(os.path.join(path, name))
            return
    for name in ("model.safetensors", "pytorch_model.bin"):
        f = os.path.join(path, name)
        if os.path.exists(f):
            yield _load_file(f)
            return
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin (or their index files) under {path}")


def canonical_key(k: str) -> str:
    """Strip PEFT wrappers: 'base_model.model.X' -> 'X', '.base_layer.' -> '.'."""
    if k.startswith("base_model.model."):
        k = k[len("base_model.model."):]
    return k.replace(".base_layer.", ".")


def load_checkpoint(model: torch.nn.Module, path: str, strict: bool = True, dtype: Optional[torch.dtype] = None):
    """Copy an HF checkpoint into `model` shard by shard.  Returns (missing, unexpected).  strict: raise on either, except for keys of
    modules the checkpoint legitimately lacks when training starts from the base LLM (grounding_encoder.*, text_hidden_fcs.*, lora_*:
    the reference creates those after from_pretrained, train_joint.py:185-232)."""
    own = dict(model.state_dict())
    seen, unexpected = set(), []
    with torch.no_grad():
        for shard in iter_checkpoint_shards(path):
            for k, v in shard.items():
                ck = canonical_key(k)
                if ck not in own:
                    unexpected.append(k)
                    continue
                dst = own[ck]
                if tuple(dst.shape) != tuple(v.shape):
                    raise RuntimeError(f"shape mismatch for {ck}: checkpoint {tuple(v.shape)} vs model {tuple(dst.shape)}")
                dst.copy_(v.to(dtype or dst.dtype))
                seen.add(ck)
            del shard
    late = ("grounding_encoder.", "text_hidden_fcs.", ".lora_A.", ".lora_B.")
    missing = [k for k in own if k not in seen]
    tied = bool(getattr(getattr(model, "config", None), "tie_word_embeddings", False))   # HF omits the tied lm_head from the checkpoint (Qwen2.5-VL-3B)
    hard_missing = [k for k in missing if not any(t in k for t in late) and not (tied and k == "lm_head.weight" and "model.embed_tokens.weight" in seen)]
    if strict and (hard_missing or unexpected):
        raise RuntimeError(f"checkpoint mismatch: missing {hard_missing[:5]} (+{max(len(hard_missing) - 5, 0)}), unexpected {unexpected[:5]} (+{max(len(unexpected) - 5, 0)})")
    return missing, unexpected


def save_checkpoint(model_or_state, path: str, max_shard_bytes: int = 5 * 2**30, config=None):
    """HF layout: model-0000i-of-0000n.safetensors + model.safetensors.index.json (one file: model.safetensors), config.json."""
    from safetensors.torch import save_file

    sd = model_or_state.state_dict() if isinstance(model_or_state, torch.nn.Module) else model_or_state
    os.makedirs(path, exist_ok=True)
    shards, cur, size = [], {}, 0
    for k, v in sd.items():
        if k == "lm_head.weight" and "model.embed_tokens.weight" in sd and v.data_ptr() == sd["model.embed_tokens.weight"].data_ptr():
            continue   # tied output embedding: one tensor, stored once under embed_tokens (HF layout; safetensors refuses shared storage)
        nb = v.numel() * v.element_size()
        if cur and size + nb > max_shard_bytes:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = v.detach().to("cpu").contiguous()
        size += nb
    if cur:
        shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    else:
        weight_map, total = {}, 0
        for i, sh in enumerate(shards):
            name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(path, name), metadata={"format": "pt"})
            for k, v in sh.items():
                weight_map[k] = name
                total += v.numel() * v.element_size()
        with open(os.path.join(path, SAFE_INDEX), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
    if config is not None:
        config.save_pretrained(path)


def merge_lora_(model: torch.nn.Module):
    """In place: fold every LoRALinear into a plain Linear (PEFT merge_and_unload: W += (alpha / r) * B @ A, delta formed in fp32 and
    added in fp32, result rounded once to the weight dtype).  Returns the names merged."""
    from ..model.qwen2_5_vl import Linear
    from ..model.qwen_train import LoRALinear

    merged = []
    for name, mod in list(model.named_modules()):
        if not isinstance(mod, LoRALinear):
            continue
        with torch.no_grad():
            A = mod.lora_A["default"].weight.float()
            B = mod.lora_B["default"].weight.float()
            w = (mod.weight.float() + mod.scaling * (B @ A)).to(mod.weight.dtype)
        lin = Linear(mod.in_features, mod.out_features, bias=mod.bias is not None, device=mod.weight.device, dtype=mod.weight.dtype)
        with torch.no_grad():
            lin.weight.copy_(w)
            if mod.bias is not None:
                lin.bias.copy_(mod.bias)
        parent = model
        parts = name.split(".")
        for p in parts[:-1]:
            parent = getattr(parent, p)
        setattr(parent, parts[-1], lin)
        merged.append(name)
    return merged
