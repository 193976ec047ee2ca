"""CPU oracle (plain fp32 PyTorch) for the rectified-flow SwiGLU head.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
mingunivision/diff_loss_rf_swiglu.py on a state dict with the reference names
(`net.time_embed.mlp.0.weight`, `net.res_blocks.{i}.…`, `net.final_layer.…`).
Pinned by tests/golden/rf_*.npz (reference outputs captured by gen_golden.py).
"""
import math

import torch
import torch.nn.functional as F


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def timestep_embedding(t, dim=256, max_period=10000):
    """TimestepEmbedder.timestep_embedding (diff_loss_rf_swiglu.py:216-234): cos || sin."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def time_embed(t, sd, prefix="net.time_embed"):
    """TimestepEmbedder.forward (:236-239): Linear(256->w) SiLU Linear(w->w)."""
    h = _lin(timestep_embedding(t), sd, prefix + ".mlp.0")
    return _lin(F.silu(h), sd, prefix + ".mlp.2")


def modulate(x, shift, scale):
    """diff_loss_rf_swiglu.py:184-185"""
    return x * (1 + scale) + shift


def res_block(x, y, sd, prefix):
    """ResBlock.forward (:268-272)."""
    w = x.shape[-1]
    shift, scale, gate = _lin(F.silu(y), sd, prefix + ".adaLN_modulation.1").chunk(3, dim=-1)
    h = F.layer_norm(x, (w,), sd[prefix + ".in_ln.weight"], sd[prefix + ".in_ln.bias"], 1e-6)
    h = modulate(h, shift, scale)
    x12 = _lin(h, sd, prefix + ".mlp.w12")
    x1, x2 = x12.chunk(2, dim=-1)
    h = _lin(F.silu(x1) * x2, sd, prefix + ".mlp.w3")
    return x + gate * h


def final_layer(x, y, sd, prefix="net.final_layer"):
    """FinalLayer.forward (:288-292): LN without affine, modulate, Linear(w->out)."""
    shift, scale = _lin(F.silu(y), sd, prefix + ".adaLN_modulation.1").chunk(2, dim=-1)
    h = modulate(F.layer_norm(x, (x.shape[-1],), None, None, 1e-6), shift, scale)
    return _lin(h, sd, prefix + ".linear")


def num_res_blocks(sd):
    n = 0
    while f"net.res_blocks.{n}.in_ln.weight" in sd:
        n += 1
    return n


def net_forward(x, t, c, sd):
    """SimpleMLPAdaLN.forward (:363-385). x [B,32], t [B] in [0,1], c [B,z]."""
    h = _lin(x, sd, "net.input_proj")
    y = time_embed(t * 1000, sd) + _lin(c, sd, "net.cond_embed")
    for i in range(num_res_blocks(sd)):
        h = res_block(h, y, sd, f"net.res_blocks.{i}")
    return final_layer(h, y, sd)


def sample(z, noise, sd, steps=16, temperature=1.0, text_cfg=3.0, image_cfg=1.1):
    """RectifiedFlowLoss.sample (:103-181), cfg_renorm_type=None and
    time_shifting_factor=None as called from modeling_bailing_moe.py:1859-1860.

    z [rows, zc]; noise [1, 32] (the reference draws it with torch.randn,
    :117-122; the oracle takes it as an argument so runs can be replayed).
    rows == 3: [cond, uncond, text_uncond];  rows == 2: [cond, uncond].
    Returns x [rows, 32] (rows identical when text_cfg != 1).
    """
    b = z.shape[0]
    if text_cfg != 1.0:
        x = torch.cat([noise] * b, dim=0) * temperature
    else:
        x = noise * temperature
        assert x.shape[0] == b
    time_steps = torch.linspace(1.0, 0.0, steps + 1)[:-1]
    step_size = 1.0 / steps
    for t in time_steps:
        t_batch = torch.ones(b) * t
        if b == 3:
            half = x[: b // 3]
            v_all = net_forward(torch.cat([half, half, half], 0), t_batch, z, sd)
            v_c, v_u, v_tu = torch.split(v_all, b // 3, dim=0)
            v = v_u + image_cfg * (v_tu - v_u) + text_cfg * (v_c - v_tu)
            v = torch.cat([v, v, v], 0)
        elif b == 2:
            half = x[: b // 2]
            v_all = net_forward(torch.cat([half, half], 0), t_batch, z, sd)
            v_c, v_u = torch.split(v_all, b // 2, dim=0)
            v = v_u + text_cfg * (v_c - v_u)
            v = torch.cat([v, v], 0)
        else:
            v = net_forward(x, t_batch, z, sd)
        x = x + v * step_size
    return x


def vis_head(h, sd, prefix="vis_head"):
    """vis_head = Linear(hidden->z) + LayerNorm(z, eps=1e-6) (modeling_bailing_moe.py:1571-1574)."""
    z = _lin(h, sd, prefix + ".0")
    return F.layer_norm(z, (z.shape[-1],), sd[prefix + ".1.weight"], sd[prefix + ".1.bias"], 1e-6)
