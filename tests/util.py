"""Shared helpers for the test-suite: golden fixtures + synthetic state dicts."""
import json
import os

import numpy as np
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in "US" and v.ndim == 0:
            out[k] = json.loads(str(v))
        elif v.ndim == 0:
            out[k] = v.item()
        else:
            out[k] = torch.from_numpy(v)
    return out


def checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def mingtok_sd(cfg_dict, seed, device="cpu", dtype=torch.float32):
    return synth_state_dict(C.mingtok_param_shapes(C.MingTokConfig(**cfg_dict)), seed, device, dtype)


def llm_sd(llm_dict, rf_dict, seed, device="cpu", dtype=torch.float32):
    return synth_state_dict(C.llm_param_shapes(C.BailingMoeConfig(**llm_dict), rf_dict, 32), seed, device, dtype)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
