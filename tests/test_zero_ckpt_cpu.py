"""ZeRO-2 checkpoint directory import (rga3.utils.zero_ckpt; reference train_joint.py:426-461 writes it, merge.sh + merge_lora_weights_and_save_hf_model.py:41-136
consume the consolidated form).  DeepSpeed is absent: reader and fixture writer restate the published 0.16.3 layout and are tested against each other (unpinned)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rga3-release_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

from make_zero2_fixture import write_zero2_checkpoint  # noqa: E402
from rga3.utils import zero_ckpt  # noqa: E402
from rga3.utils.checkpoint import load_checkpoint  # noqa: E402


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = torch.nn.Embedding(37, 12)
        self.frozen = torch.nn.Linear(12, 20)
        self.lora_A = torch.nn.Linear(12, 4, bias=False)
        self.lora_B = torch.nn.Linear(4, 20, bias=False)
        self.head = torch.nn.Linear(20, 37, bias=False)
        self.register_buffer("pos", torch.arange(5, dtype=torch.float32))


def _make(tmp_path, world, prefix=""):
    torch.manual_seed(3)
    m = Tiny()
    for p in m.frozen.parameters():
        p.requires_grad_(False)
    groups = [["lora_A.weight", "lora_B.weight", "head.weight"], ["embed.weight"]]     # two optimizer groups (e.g. different lr / decay)
    master = {n: p.detach().float() + 1e-3 * torch.randn_like(p) for n, p in m.named_parameters() if p.requires_grad}   # masters differ from the bf16 copies
    if prefix:       # the reference wraps the model in PEFT before handing it to the engine: names carry base_model.model.
        class Wrap(torch.nn.Module):
            def __init__(self, inner):
                super().__init__()
                self.base_model = torch.nn.Module()
                self.base_model.model = inner
        wm = Wrap(m)
        groups = [["base_model.model." + n for n in g] for g in groups]
        master = {"base_model.model." + n: v for n, v in master.items()}
        write_zero2_checkpoint(wm, master, groups, str(tmp_path), world)
    else:
        write_zero2_checkpoint(m, master, groups, str(tmp_path), world)
    return m, master


@pytest.mark.parametrize("world", [1, 2, 8])
def test_consolidate_zero2_round_trip(tmp_path, world):
    m, master = _make(tmp_path, world)
    sd = zero_ckpt.consolidate_zero2(str(tmp_path))
    assert set(sd) == set(m.state_dict())
    for n, v in master.items():
        assert sd[n].dtype == torch.float32 and torch.equal(sd[n], v), n           # the fp32 MASTER weights, bit for bit, not the bf16 module copies
    assert torch.equal(sd["frozen.weight"], m.frozen.weight.detach().float()) and torch.equal(sd["pos"], m.pos)
    assert set(zero_ckpt.consolidate_zero2(str(tmp_path), exclude_frozen=True)) == set(master) | {"pos"}


def test_zero_to_fp32_shards_load_like_the_reference_merge_script(tmp_path):
    """merge.sh: zero_to_fp32.py <ckpt> <out>; merge_lora_weights_and_save_hf_model.py:124-131 then reads the index + shards.  Names carry PEFT's prefix."""
    m, master = _make(tmp_path / "ckpt", 4, prefix="peft")
    out = zero_ckpt.zero_to_fp32(str(tmp_path / "ckpt"), str(tmp_path / "pytorch_model"), max_shard_bytes=2048)
    files = sorted(os.listdir(out))
    assert "pytorch_model.bin.index.json" in files and sum(f.endswith(".bin") for f in files) >= 2
    fresh = Tiny()
    load_checkpoint(fresh, out, strict=True)
    for n, v in master.items():
        assert torch.equal(dict(fresh.named_parameters())[n[len("base_model.model."):]].detach(), v)
    fresh2 = Tiny()
    missing, unexpected = zero_ckpt.load_zero_checkpoint(fresh2, str(tmp_path / "ckpt"))
    assert not missing and not unexpected
    assert torch.equal(fresh2.head.weight.detach(), master["base_model.model.head.weight"])


def test_rejects_what_it_cannot_consolidate(tmp_path):
    m, _ = _make(tmp_path, 2)
    d = os.path.join(str(tmp_path), "global_step7")
    f1 = os.path.join(d, "bf16_zero_pp_rank_1_mp_rank_00_optim_states.pt")
    o = torch.load(f1, weights_only=False)
    o["optimizer_state_dict"]["zero_stage"] = 3
    torch.save(o, f1)
    with pytest.raises(ValueError, match="stage 3"):
        zero_ckpt.consolidate_zero2(str(tmp_path))
    os.remove(f1)
    with pytest.raises(ValueError, match="partition_count"):
        zero_ckpt.consolidate_zero2(str(tmp_path))
    os.remove(os.path.join(str(tmp_path), "latest"))
    with pytest.raises(FileNotFoundError):
        zero_ckpt.consolidate_zero2(str(tmp_path))
    assert zero_ckpt.consolidate_zero2.__doc__
