"""Launch script for tools/pmc_kernel.sh: the RF sampler at the reference's call shape (2 CFG rows, full 16B-A3B head) — the whole sampler is
ONE persistent launch (`rf_blocks_persist_kernel`, stream_kc.hip).  usage: prof_rf_persist.py [bf16|fp8|int4] [calls]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
weights = sys.argv[1] if len(sys.argv) > 1 else "bf16"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
args = argparse.Namespace(tiny=False, tokens=256, layers=2, prompt_len=40, images=1, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
if weights != "bf16":
    rf = rf.to_fp8(weights)
g = torch.Generator(device=dev).manual_seed(0)
hid = torch.randn(2, cfg.hidden_size, device=dev, generator=g)
noise = torch.randn(1, 32, device=dev, generator=g)
for _ in range(calls):
    lat = rf.sample(hid, noise, n_images=1)
torch.cuda.synchronize()
assert torch.isfinite(lat).all()
print("ok", lat.flatten()[:4].tolist())
