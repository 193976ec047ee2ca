"""Deterministic synthetic weights, keyed by parameter name.

The real Ming-UniVision / MingTok-Vision checkpoints are not available offline
(SURVEY.md "Read this first" item 5), so benchmarks and parity tests run on
random-init weights of the exact architecture.  Every tensor is drawn from a
generator seeded by crc32(parameter name) so that any shard / layer can be
produced independently (and on any rank) without shipping 34 GB around.

Rule (`init_rule`): 1-D `*.weight` (norm gains) ~ 1 + N(0, 0.02); biases ~
N(0, 0.02); cls/pos tokens ~ N(0, 0.02); everything else ~ N(0, 1/sqrt(fan_in))
which keeps activations O(1) through 28/24/12 layers (close to the reference's
N(0, 0.02) at fan_in 2048-3072).  The normally zero-initialised RF-head layers
(diff_loss_rf_swiglu.py:352-361) are drawn like any other Linear so that the head
is not the identity.  Values are rounded to bf16 (the storage type of the
weights in HBM) and returned in the requested dtype.
"""
import math
import zlib

import torch


def _seed(name: str, base_seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (base_seed * 0x9E3779B1)) & 0x7FFFFFFF


def init_rule(name: str, shape):
    """-> (mean, std) for the parameter called `name`."""
    if name.endswith("cls_token") or name.endswith("pos_embed"):
        return 0.0, 0.02
    if name.endswith(".bias"):
        return 0.0, 0.02
    if len(shape) == 1:
        return 1.0, 0.02
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return 0.0, 1.0 / math.sqrt(fan_in)


def synth_tensor(name, shape, base_seed=0, device="cpu", dtype=torch.float32, mean=None, std=None):
    m, s = init_rule(name, shape)
    mean = m if mean is None else mean
    std = s if std is None else std
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(_seed(name, base_seed))
    t = torch.empty(tuple(shape), dtype=torch.float32, device=dev)
    t.normal_(mean, std, generator=g)
    return t.to(torch.bfloat16).to(dtype)


def synth_state_dict(shapes: dict, base_seed=0, device="cpu", dtype=torch.float32):
    """shapes: {param name: shape}. Returns {name: tensor}."""
    return {k: synth_tensor(k, v, base_seed, device, dtype) for k, v in shapes.items()}
