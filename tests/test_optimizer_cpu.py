"""Host logic of the optimizer side of the training step: DeepSpeed WarmupCosineLR as configured by reference train_joint.py:308-317 (restated from the
published lr_schedules.py; DeepSpeed is not installed: parity unpinned) and FusedAdamW's resume state (optimizer / scheduler state must
survive a checkpoint the way the reference's auto_resume does, train_joint.py:352-366)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))

from rga3.parallel.ddp import FusedAdamW, WarmupCosineLR  # noqa: E402


def test_warmup_cosine_schedule_values():
    total = 80 * 100                                   # run_torchrun.sh: 80 epochs x 100 steps
    s = WarmupCosineLR(total, int(0.03 * total), warmup_min_ratio=0.0, cos_min_ratio=0.03, warmup_type="linear")
    assert s.warm == 240
    assert s.scale_at(0) == 1.0                        # the engine steps the scheduler after the optimizer: step 0 runs at the configured lr
    assert s.scale_at(1) == 0.0                        # last_batch_iteration 0: 0 / warmup
    assert abs(s.scale_at(121) - 120 / 240) < 1e-12
    assert abs(s.scale_at(241) - (0.03 + 0.97 * (1 + math.cos(math.pi * 1 / (total - 240))) / 2)) < 1e-12
    assert abs(s.ratio(total - 1) - 0.03) < 1e-9       # cosine floor at the end
    prev = 2.0
    for it in range(240, total, 97):                   # monotone decay after the warm-up
        r = s.ratio(it)
        assert r <= prev + 1e-12
        prev = r


def test_fused_adamw_state_round_trip():
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 8).bfloat16()), torch.nn.Parameter(torch.randn(8).bfloat16())]
    sch = WarmupCosineLR(1000, 30)
    a = FusedAdamW(ps, lr=4e-5, schedule=sch)
    a.t = 7
    for x in a.m + a.v:
        x.normal_()
    for w in a.master:
        w.add_(1e-3)                                   # masters carry bits the bf16 parameters do not
    sd = a.state_dict()
    qs = [torch.nn.Parameter(torch.zeros(5, 8).bfloat16()), torch.nn.Parameter(torch.zeros(8).bfloat16())]
    b = FusedAdamW(qs, lr=1.0)
    b.load_state_dict(sd)
    assert b.t == 7 and b.lr == 4e-5 and b.schedule.total == 1000 and b.schedule.warm == 30
    assert abs(b.current_lr() - 4e-5 * sch.scale_at(7)) < 1e-18
    for x, y in zip(a.master + a.m + a.v, b.master + b.m + b.v):
        assert torch.equal(x, y)
    for q, w in zip(qs, a.master):                     # parameters restored as the rounded masters
        assert torch.equal(q.data, w.to(torch.bfloat16))
    # a checkpoint loaded into the model AFTER the optimizer was built must not be overwritten by stale masters
    with torch.no_grad():
        qs[0].fill_(0.5)
    b.resync_master()
    assert float(b.master[0].min()) == 0.5
