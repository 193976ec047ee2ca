"""Wide-row RF sampler (rows > 64, gemm256 route) against the <= 64-row route of the same library, image by image,
at full width (w = 3072, depth 12, hidden 8192, 16 steps) + timing at 256 / 512 / 1024 rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.rf_head import RectifiedFlowHead
from ming_univision_amd.synth import synth_tensor
dev = "cuda"
cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
full = C.llm_param_shapes(cfg, rf_cfg, 32)
rf_sd = {k: synth_tensor(k, s, 7, dev, torch.bfloat16) for k, s in full.items() if k.startswith("vis_head") or k.startswith("diffloss")}
rf = RectifiedFlowHead(rf_sd, cfg.hidden_size, rf_cfg)
g = torch.Generator().manual_seed(3)
B = 40
hidden = torch.randn(2 * B, cfg.hidden_size, generator=g).to(dev)
noise = torch.randn(B, 32, generator=g).to(dev)
wide = rf.sample(hidden, noise, n_images=B)
nar = torch.cat([rf.sample(hidden[2 * i:2 * i + 64].contiguous(), noise[i:i + 32].contiguous(), n_images=min(32, B - i)) for i in range(0, B, 32)])
torch.cuda.synchronize()
err = ((wide - nar).abs().amax(dim=1) / nar.abs().amax(dim=1))
print(f"wide (80 rows) vs narrow route, per-image rel err: max {float(err.max()):.2e} median {float(err.median()):.2e}; finite {bool(torch.isfinite(wide).all())}", flush=True)
for Bn in (128, 256, 512):
    hidden = torch.randn(2 * Bn, cfg.hidden_size, generator=g).to(dev)
    noise = torch.randn(Bn, 32, generator=g).to(dev)
    out = rf.sample(hidden, noise, n_images=Bn)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): rf.sample(hidden, noise, n_images=Bn, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 3
    print(f"rows {2*Bn}: {ms:.2f} ms per token -> RF-only {Bn / ms * 1e3:.0f} image-tokens/s; finite {bool(torch.isfinite(out).all())}", flush=True)
