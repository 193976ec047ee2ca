"""A few lock-step visual tokens at the bench's operating point (768 images = 1536 rows) with a short LLM stack, for rocprofv3
passes over the wide route's launches (--kernel-trace --stats, or one --pmc counter at a time): the expert GEMMs' fabric traffic.
usage: prof_wide_tokens.py [images] [layers] [tokens (a square: the pixel decoder runs)]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tokens = int(sys.argv[3]) if len(sys.argv) > 3 else 4
args = argparse.Namespace(tiny=False, tokens=tokens, layers=layers, prompt_len=40, images=B, cfg_rows=2)
dev = torch.device("cuda")
cfg, dec, rf, tok = bench.build_models(args, dev, seed=0)
g = torch.Generator(device=dev).manual_seed(0)
prompt = torch.randint(0, 100000, (B, 40), generator=g, device=dev)
noises = torch.randn(B, tokens + 1, 32, generator=g, device=dev)
bench.one_image(cfg, dec, rf, tok, prompt, noises)
torch.cuda.synchronize()
