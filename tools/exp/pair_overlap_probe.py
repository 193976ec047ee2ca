"""Does the (row, expert) pair route already profit from rows that select the SAME experts (both pairs' workgroups stream the same matrix at
the same time: L2 / Infinity Cache)?  Decoder step at 2 rows, full 16B-A3B stack: random rows (about 11 distinct routed experts of 12
pairs per layer) against two identical rows (6 distinct)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=1, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
g = torch.Generator(device=dev).manual_seed(1)
small = dec.view(t_max=200, n_seq=4)
for rows in (2, 3):
    x = torch.randn(rows, cfg.hidden_size, device=dev, generator=g)
    xs = x[:1].expand(rows, -1).contiguous()
    seq = torch.arange(rows, dtype=torch.int32, device=dev); slot = torch.full((rows,), 60, dtype=torch.int32, device=dev)
    out = {}
    for rnd in range(3):
        for name, xx in (("random rows", x), ("identical rows", xs)):
            f = lambda: small.step(xx, seq, slot, slot, slot + 1, distinct_sequences=True)
            f(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): f()
            e.record(); torch.cuda.synchronize()
            out.setdefault(name, []).append(s.elapsed_time(e) / 20)
    print(f"{rows} rows: " + ";  ".join(f"{k} {min(v):.3f} ms ({', '.join('%.3f' % t for t in v)})" for k, v in out.items()), flush=True)
