"""One image at batch 1 (the reference's call shape: 2 CFG rows), a few visual tokens of the full 16B-A3B path, for rocprofv3
--kernel-trace + tools/site_stats.py: the launch sequence of one visual token by call site."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse
import torch
import bench

args = argparse.Namespace(tiny=False, tokens=int(sys.argv[1]) if len(sys.argv) > 1 else 8, layers=None, prompt_len=40, images=1, cfg_rows=2)
dev = torch.device("cuda")
cfg, dec, rf, tok = bench.build_models(args, dev, seed=0)
g = torch.Generator(device=dev).manual_seed(0)
prompt = torch.randint(0, 100000, (1, 40), generator=g, device=dev)
noises = torch.randn(1, args.tokens + 1, 32, generator=g, device=dev)
bench.one_image(cfg, dec, rf, tok, prompt, noises)
torch.cuda.synchronize()
