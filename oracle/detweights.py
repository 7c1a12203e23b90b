"""ORACLE support — deterministic, name-keyed pseudo-random tensors.

Fixtures store only inputs/outputs: every weight is regenerated bit-identically on any machine from its
parameter name (numpy PCG64 seeded by crc32(name)), so the same tensors can be poured into the reference's
modules (fixture generation, this container only) and into the oracle / product modules (tests, GPU box).
"""
import zlib

import numpy as np
import torch


def det_array(name: str, shape, scale: float = 1.0, seed: int = 0, offset: float = 0.0) -> np.ndarray:
    rng = np.random.default_rng((zlib.crc32(name.encode()) + 7919 * seed) & 0xFFFFFFFF)
    return (rng.standard_normal(tuple(shape)) * scale + offset).astype(np.float32)


def det_tensor(name: str, shape, scale: float = 1.0, seed: int = 0, offset: float = 0.0) -> torch.Tensor:
    return torch.from_numpy(det_array(name, shape, scale, seed, offset))


def det_state_dict(shapes: dict, seed: int = 0, rules=None) -> dict:
    """shapes: name -> shape.  Default rule: norm-like 1-D '.weight' of norms -> 1 + 0.1 n; biases 0.02 n;
    matrices n / sqrt(fan_in)."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if rules is not None:
            r = rules(name, shape)
            if r is not None:
                out[name] = det_tensor(name, shape, r[0], seed, r[1])
                continue
        if len(shape) <= 1:
            if name.endswith("bias"):
                out[name] = det_tensor(name, shape, 0.02, seed)
            elif "norm" in name or "ln_" in name or name.endswith("g_weight"):
                out[name] = det_tensor(name, shape, 0.1, seed, 1.0)
            else:
                out[name] = det_tensor(name, shape, 0.5, seed)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[name] = det_tensor(name, shape, 1.0 / np.sqrt(max(fan_in, 1)), seed)
    return out
