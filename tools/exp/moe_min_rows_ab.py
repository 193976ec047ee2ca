"""A/B of the first row count of a decoder step that takes the grouped expert route (sorted pairs, streaming MFMA launches, combine glue)
instead of the (row, expert) pair launches on the fp32-FMA kernels + the wave-segmented down projection: ms per decoder step on the
full 28-layer 16B-A3B stack at 2-6 rows, interleaved arms, and the difference of the hidden states.
    python tools/exp/moe_min_rows_ab.py [bf16|fp8]"""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_moe_tune_min_rows.argtypes = [ctypes.c_int]; L.mn_moe_tune_min_rows.restype = None
dev = torch.device("cuda", 0)
weights = sys.argv[1] if len(sys.argv) > 1 else "bf16"
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=1, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
if weights != "bf16":
    dec = dec.to_fp8(n_seq=8, weights=weights)
g = torch.Generator(device=dev).manual_seed(1)
L.mn_moe_tune_min_rows(3)                          # (workspace sized for the grouped route)
small = dec.view(t_max=200, n_seq=8)
for rows in (3, 4, 5, 6):
    x = torch.randn(rows, cfg.hidden_size, device=dev, generator=g)
    seq = torch.arange(rows, dtype=torch.int32, device=dev); slot = torch.full((rows,), 60, dtype=torch.int32, device=dev)
    def run(): return small.step(x, seq, slot, slot, slot + 1, distinct_sequences=True)
    res = {}
    for rnd in range(3):
        for arm in (3, 64):
            L.mn_moe_tune_min_rows(arm)
            run(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): out = run()
            e.record(); torch.cuda.synchronize()
            res.setdefault(arm, []).append((s.elapsed_time(e) / 20, out.clone()))
    a, b = res[3], res[64]
    d = ((a[0][1] - b[0][1]).abs().max() / a[0][1].abs().max()).item()
    print(f"{weights} {rows} rows: grouped route {min(t for t, _ in a):.3f} ms ({', '.join('%.3f' % t for t, _ in a)});  pair launches "
          f"{min(t for t, _ in b):.3f} ms ({', '.join('%.3f' % t for t, _ in b)});  hidden states differ by {d:.1e}, finite {bool(torch.isfinite(b[0][1]).all())}", flush=True)
L.mn_moe_tune_min_rows(0)
