"""Writes a DeepSpeed ZeRO stage-2 checkpoint directory for a given module, in the layout `model_engine.save_checkpoint` leaves (reference train_joint.py:426-461) as
described by the published DeepSpeed 0.16.3 sources (engine.py `_save_zero_checkpoint`, `stage_1_and_2.py` `state_dict`, `utils/zero_to_fp32.py`).  DeepSpeed is not
in the image, so this is a restatement of the WRITER; rga3.utils.zero_ckpt restates the READER -- the two are tested against each other, parity with a directory a real
DeepSpeed run wrote is unpinned.

    write_zero2_checkpoint(module, fp32_master, groups, out_dir, world, tag="global_step7", bf16=True)

  module        the engine's module (its bf16 state dict goes to mp_rank_00_model_states.pt)
  fp32_master   {name: fp32 tensor} of the TRAINABLE parameters (the optimizer's master weights: what consolidation must return)
  groups        list of lists of trainable parameter names, one list per optimizer param group, in flattening order
"""
import math
import os
from collections import OrderedDict

import torch


def write_zero2_checkpoint(module, fp32_master, groups, out_dir, world, tag="global_step7", bf16=True, shared=()):
    d = os.path.join(out_dir, tag)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(out_dir, "latest"), "w") as f:
        f.write(tag)
    named = dict(module.named_parameters())
    trainable = {n for g in groups for n in g}
    buffer_names = [n for n, _ in module.named_buffers()]
    frozen = OrderedDict((n, p) for n, p in named.items() if n not in trainable)
    model_states = {
        "module": {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in module.state_dict().items()},
        "buffer_names": buffer_names,
        "param_shapes": [OrderedDict((n, named[n].shape) for n in g) for g in groups],
        "frozen_param_shapes": OrderedDict((n, p.shape) for n, p in frozen.items()) or None,
        "frozen_param_fragments": OrderedDict((n, p.detach().float().reshape(-1).clone()) for n, p in frozen.items()) or None,
        "shared_params": [list(s) for s in shared],
        "ds_version": "0.16.3",
    }
    torch.save(model_states, os.path.join(d, "mp_rank_00_model_states.pt"))
    # flat fp32 groups, padded to a multiple of 2 * world (DeepSpeed's flattening alignment), cut into equal rank partitions
    flats = []
    for g in groups:
        flat = torch.cat([fp32_master[n].float().reshape(-1) for n in g])
        align = 2 * world
        pad = align * math.ceil(flat.numel() / align) - flat.numel()
        flats.append(torch.cat([flat, torch.zeros(pad)]))
    for r in range(world):
        parts = [f.view(world, -1)[r].clone() for f in flats]
        osd = {"optimizer_state_dict": {"zero_stage": 2, "partition_count": [world] * len(groups), "single_partition_of_fp32_groups": parts,
                                        "base_optimizer_state": {}, "loss_scaler": None, "overflow": False},
               "ds_config": {}, "ds_version": "0.16.3"}
        name = ("bf16_" if bf16 else "") + f"zero_pp_rank_{r}_mp_rank_00_optim_states.pt"
        torch.save(osd, os.path.join(d, name))
    return d
