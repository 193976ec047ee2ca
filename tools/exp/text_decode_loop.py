"""One-row text decode (bench.py's `text_decode` leg: 40-token prompt, 64 new tokens, ids and bookkeeping on the device) as a
stand-alone loop — meant to run under rocprofv3 --kernel-trace (tools/prof_textdecode_sites.sh).  argv[1]: bf16 | fp8 | int8."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "bf16"
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=1, cfg_rows=2, weights=w)
dev = torch.device("cuda", 0)
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
small = dec.view(t_max=args.prompt_len + 80, n_seq=1)
prompt = torch.randint(0, 100000, (args.prompt_len,), device=dev)
for _ in range(3):
    print("%.1f tokens/s" % bench.text_decode_rate(small, prompt), flush=True)
