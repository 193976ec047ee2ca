"""What fp8 (OCP e4m3fn) WEIGHTS would cost in parity on the HBM-bound path (BASELINE configs[4]'s fp8 clause).
The RF head at full width, 2 CFG rows (<= 64-row route): w12 / w3 / adaLN weights rounded to e4m3 with a per-output-row scale (the
best case for a weight-only fp8 scheme: dequantised on load, fp32-class arithmetic otherwise), sampled latents against the fp32
oracle with the ORIGINAL weights.  No fp8 kernel is needed for the error: the rounded values are exactly representable in bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.rf_head import RectifiedFlowHead
from ming_univision_amd.synth import synth_state_dict
from oracle import rf_ref
from tests.util import rel_err

cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
shapes = {k: s for k, s in C.llm_param_shapes(cfg, rf_cfg, 32).items() if k.startswith("vis_head") or k.startswith("diffloss")}
sd = {k: v.to(torch.bfloat16) for k, v in synth_state_dict(shapes, 7).items()}
torch.set_num_threads(min(32, torch.get_num_threads()))


def fp8_round(w):
    s = w.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0
    return ((w.float() / s).to(torch.float8_e4m3fn).float() * s).to(torch.bfloat16)


g = torch.Generator().manual_seed(0)
hidden = torch.randn(2, cfg.hidden_size, generator=g)
noise = torch.randn(1, 32, generator=g)
sd32 = {k: v.float() for k, v in sd.items()}
z = rf_ref.vis_head(hidden, sd32)
ref = rf_ref.sample(z, noise, {k[len("diffloss."):]: v for k, v in sd32.items() if k.startswith("diffloss.")}, steps=16)


def run(sdq):
    rf = RectifiedFlowHead({k: v.cuda().contiguous() for k, v in sdq.items()}, cfg.hidden_size, rf_cfg)
    return rf.sample(hidden.cuda(), noise.cuda()[0], n_images=1)


print("bf16 weights (shipped):            latent rel err vs fp32 oracle %.2e" % rel_err(run(sd), ref[0]))
for tag, pick in (("w12 + w3 in e4m3", lambda k: ".mlp.w12.weight" in k or ".mlp.w3.weight" in k),
                  ("w12 + w3 + adaLN in e4m3", lambda k: ".mlp.w" in k and k.endswith("weight") or "adaLN_modulation.1.weight" in k)):
    q = {k: (fp8_round(v) if pick(k) and v.dim() == 2 else v) for k, v in sd.items()}
    print("%-34s latent rel err vs fp32 oracle %.2e   (bar 1e-3)" % (tag + ":", rel_err(run(q), ref[0])))
