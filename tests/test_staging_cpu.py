"""Host logic of rga3.utils.staging: the CPU companion of a moved tensor is trusted only while neither side has been written (version counters), and
dict_to_cuda keeps the reference's semantics (utils/utils.py:177-187: tensors and lists of tensors move, everything else stays)."""
import numpy as np
import torch

from rga3.utils import staging


def test_host_of_plain_cases():
    assert staging.host_of(None) is None
    a = np.arange(4)
    assert staging.host_of(a) is a
    t = torch.arange(6).view(2, 3)
    assert np.array_equal(staging.host_of(t), t.numpy())
    assert np.array_equal(staging.host_of([1, 2]), np.array([1, 2]))


def test_companion_is_dropped_after_a_write():
    cpu = torch.arange(8)
    dev = cpu.clone()                      # stands in for the device copy (the attribute logic does not depend on the device type)
    staging.attach_host(dev, cpu)
    assert staging.has_host(dev)
    cpu.add_(1)                            # the collate buffer was refilled in place: the companion no longer describes `dev`
    assert not staging.has_host(dev)
    staging.attach_host(dev, cpu)
    dev.mul_(2)                            # or the device tensor was written
    assert not staging.has_host(dev)


def test_dict_to_cuda_semantics_on_cpu():
    d = {"ids": torch.arange(4), "px": torch.randn(3, 2), "lst": [torch.ones(2), torch.zeros(2)], "flag": False, "sizes": [(1, 2)], "empty": []}
    out = staging.dict_to_cuda(dict(d), device="cpu")
    assert set(out) == set(d) and out["flag"] is False and out["sizes"] == [(1, 2)] and out["empty"] == []
    assert torch.equal(out["ids"], d["ids"]) and torch.equal(out["lst"][1], d["lst"][1])


def test_upload_cpu_passthrough():
    a = np.arange(5, dtype=np.int64)
    t = staging.upload(a, "cpu")
    assert t.dtype == torch.int64 and np.array_equal(t.numpy(), a)
    assert staging.upload(np.zeros(0, dtype=np.int32), "cpu").numel() == 0
