"""DeepSpeed ZeRO-2 checkpoint directory -> fp32 state dict (VERDICT r3 missing item 4).

The reference trainer saves `model_engine.save_checkpoint(<log_dir>/ckpt_latest)` (reference train_joint.py:426-461) and later runs DeepSpeed's
`zero_to_fp32.py <ckpt_dir> <out_dir>` (reference merge.sh) before `merge_lora_weights_and_save_hf_model.py:41-136` loads the consolidated shards.  This build's
own DDP optimizer keeps whole parameters on every rank and never writes this format -- but a user switching over has such directories.

DeepSpeed (pinned 0.16.3, reference requirements.txt:2) is not in the image and the reference ships no checkpoint: the layout and the merge below are RESTATED
from the published `deepspeed/utils/zero_to_fp32.py` (parity unpinned; the fixture of tests/test_zero_ckpt_cpu.py is written by
tests/golden/make_zero2_fixture.py from the same description):

  <dir>/latest                                         text file: the tag, e.g. "global_step1200"
  <dir>/<tag>/mp_rank_00_model_states.pt               {"module": bf16 state dict (all params + buffers), "buffer_names": [...],
                                                        "param_shapes": [ {name: torch.Size} per optimizer group, in flattening order ],
                                                        "frozen_param_shapes": {name: Size} | None, "frozen_param_fragments": {name: tensor} | None,
                                                        "shared_params": [[alias, source], ...], "ds_version": str}
  <dir>/<tag>/[bf16_]zero_pp_rank_<r>_mp_rank_00_optim_states.pt
                                                       {"optimizer_state_dict": {"zero_stage": 1 | 2, "partition_count": int | [int],
                                                        "single_partition_of_fp32_groups": [flat fp32 partition of each group]}}

Stage <= 2 merge: per optimizer group, the ranks' partitions are concatenated in rank order into the flat fp32 group; the parameters are cut out of it in
`param_shapes` order; the group is padded to a multiple of 2 * world_size elements (the alignment DeepSpeed flattens with), which is checked.  Frozen parameters
come whole from `frozen_param_fragments`, buffers from the module state dict, shared parameters are aliased.  Names are the engine's module names (PEFT-wrapped in
the reference: `base_model.model.` ...), which rga3.utils.checkpoint.load_checkpoint accepts."""
from __future__ import annotations

import glob
import json
import math
import os
import re
from typing import Dict, Optional

import torch

MODEL_STATES = "mp_rank_00_model_states.pt"


def _natural(s: str):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


def _tag_dir(ckpt_dir: str, tag: Optional[str]) -> str:
    if tag is None:
        latest = os.path.join(ckpt_dir, "latest")
        if not os.path.isfile(latest):
            raise FileNotFoundError(f"{latest}: a DeepSpeed checkpoint directory carries a 'latest' file naming its tag (pass tag= otherwise)")
        with open(latest) as f:
            tag = f.read().strip()
    d = os.path.join(ckpt_dir, tag)
    if not os.path.isdir(d):
        raise FileNotFoundError(f"{d}: checkpoint tag directory not found")
    return d


def consolidate_zero2(ckpt_dir: str, tag: Optional[str] = None, exclude_frozen: bool = False) -> Dict[str, torch.Tensor]:
    """fp32 state dict of a ZeRO stage-1/2 checkpoint directory (what `zero_to_fp32.py` reconstructs)."""
    d = _tag_dir(ckpt_dir, tag)
    ms_path = os.path.join(d, MODEL_STATES)
    if not os.path.isfile(ms_path):
        raise FileNotFoundError(f"{ms_path} not found (model-parallel checkpoints, mp_rank > 0, are not produced by the reference trainer)")
    ms = torch.load(ms_path, map_location="cpu", weights_only=False)
    optim_files = sorted(glob.glob(os.path.join(d, "*_optim_states.pt")), key=_natural)
    if not optim_files:
        raise FileNotFoundError(f"no *_optim_states.pt under {d}")
    parts, stage, world = [], None, None
    for f in optim_files:
        osd = torch.load(f, map_location="cpu", weights_only=False)["optimizer_state_dict"]
        st = int(osd["zero_stage"])
        if st > 2:
            raise ValueError(f"{f}: ZeRO stage {st}: only stages 1 and 2 (what the reference trains with, train_joint.py:325-346) are consolidated here")
        pc = osd["partition_count"]
        pc = max(pc) if isinstance(pc, (list, tuple)) else int(pc)
        if stage is None:
            stage, world = st, pc
        if (st, pc) != (stage, world):
            raise ValueError(f"{f}: stage / partition count ({st}, {pc}) differs from the first rank's ({stage}, {world})")
        parts.append([t.float().reshape(-1) for t in osd["single_partition_of_fp32_groups"]])
    if world != len(optim_files):
        raise ValueError(f"{d}: partition_count {world} but {len(optim_files)} optimizer files")
    param_shapes = ms["param_shapes"]
    if len(param_shapes) != len(parts[0]):
        raise ValueError(f"{d}: {len(param_shapes)} parameter groups in the model states, {len(parts[0])} in the optimizer states")
    sd: Dict[str, torch.Tensor] = {}
    # buffers (as fp32, like zero_to_fp32)
    for name in ms.get("buffer_names", []) or []:
        sd[name] = ms["module"][name].float()
    # frozen parameters: stored whole by stage <= 2
    if not exclude_frozen and ms.get("frozen_param_shapes"):
        frag = ms.get("frozen_param_fragments") or {}
        for name, shape in ms["frozen_param_shapes"].items():
            if name not in frag:
                raise KeyError(f"frozen parameter {name} has a shape entry but no fragment")
            t = frag[name].float()
            if t.numel() != math.prod(shape):
                raise ValueError(f"frozen parameter {name}: fragment has {t.numel()} elements, shape {tuple(shape)}")
            sd[name] = t.reshape(tuple(shape))
    # trainable parameters: cut out of the concatenated fp32 groups
    align = 2 * world
    for gi, shapes in enumerate(param_shapes):
        full = torch.cat([p[gi] for p in parts])
        off = 0
        for name, shape in shapes.items():
            n = math.prod(shape)
            if off + n > full.numel():
                raise ValueError(f"group {gi}: parameter {name} ends at {off + n}, the merged group has {full.numel()} elements")
            sd[name] = full.narrow(0, off, n).reshape(tuple(shape)).clone()
            off += n
        if align * math.ceil(off / align) != align * math.ceil(full.numel() / align):
            raise ValueError(f"group {gi}: consumed {off} of {full.numel()} elements (alignment {align}): parameter list and partitions disagree")
    for pair in ms.get("shared_params", []) or []:
        if pair[1] in sd:
            sd[pair[0]] = sd[pair[1]]
    return sd


def zero_to_fp32(ckpt_dir: str, out_dir: str, tag: Optional[str] = None, max_shard_bytes: int = 5 * 2 ** 30) -> str:
    """`python zero_to_fp32.py <ckpt_dir> <out_dir>` of reference merge.sh: consolidated fp32 weights as sharded `pytorch_model-0000i-of-0000n.bin` +
    `pytorch_model.bin.index.json` (the layout merge_lora_weights_and_save_hf_model.py:124-131 reads)."""
    sd = consolidate_zero2(ckpt_dir, tag)
    os.makedirs(out_dir, exist_ok=True)
    shards, cur, size = [], {}, 0
    for k, v in sd.items():
        b = v.numel() * v.element_size()
        if cur and size + b > max_shard_bytes:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = v.contiguous()
        size += b
    if cur:
        shards.append(cur)
    weight_map, total = {}, 0
    for i, sh in enumerate(shards):
        name = f"pytorch_model-{i + 1:05d}-of-{len(shards):05d}.bin"
        torch.save(sh, os.path.join(out_dir, name))
        for k, v in sh.items():
            weight_map[k] = name
            total += v.numel() * v.element_size()
    with open(os.path.join(out_dir, "pytorch_model.bin.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)
    return out_dir


def load_zero_checkpoint(model: torch.nn.Module, ckpt_dir: str, tag: Optional[str] = None, strict: bool = True, dtype: Optional[torch.dtype] = None):
    """Pour a ZeRO-2 checkpoint directory into `model` (names through rga3.utils.checkpoint.canonical_key: PEFT's `base_model.model.` / `.base_layer.` wrappers
    accepted; LoRA factors land in this build's LoRALinear modules, or are merged afterwards with rga3.utils.checkpoint.merge_lora_)."""
    from .checkpoint import canonical_key

    sd = {canonical_key(k): v for k, v in consolidate_zero2(ckpt_dir, tag).items()}
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    if strict and (missing or unexpected):
        raise RuntimeError(f"ZeRO checkpoint does not match the model: missing {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                           f"unexpected {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                if tuple(own[k].shape) != tuple(v.shape):
                    raise RuntimeError(f"{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(own[k].shape)}")
                own[k].copy_(v.to(dtype or own[k].dtype))
    return missing, unexpected
