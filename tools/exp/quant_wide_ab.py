"""Weight-only modes on the wide route (65+ rows): a decoder step with layer l + 1's expert codes expanded on a side stream under layer l's
GEMMs against the in-line expansion, and the bf16 model's step beside them: ms per 28-layer step.
    python tools/exp/quant_wide_ab.py [fp8|int8|int4] [rows,...]"""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_wide_tune_dequant.argtypes = [ctypes.c_int]; L.mn_wide_tune_dequant.restype = None
dev = torch.device("cuda", 0)
weights = sys.argv[1] if len(sys.argv) > 1 else "fp8"
ROWS = [int(r) for r in sys.argv[2].split(",")] if len(sys.argv) > 2 else [66, 130, 512, 1536]
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=max(ROWS) // 2, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
del rf, tok
decq = dec.to_fp8(n_seq=max(ROWS), weights=weights)
g = torch.Generator(device=dev).manual_seed(1)
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for rows in ROWS:
    x = torch.randn(rows, cfg.hidden_size, device=dev, generator=g)
    seq = torch.arange(rows, dtype=torch.int32, device=dev); slot = torch.full((rows,), 60, dtype=torch.int32, device=dev)
    outs = {}
    t16 = ev(lambda: dec.step(x, seq, slot, slot, slot + 1, distinct_sequences=True))
    ts = {}
    for ov in (0, 1, 0, 1):
        L.mn_wide_tune_dequant(ov)
        ts.setdefault(ov, []).append(ev(lambda: decq.step(x, seq, slot, slot, slot + 1, distinct_sequences=True)))
        outs[ov] = decq.step(x, seq, slot, slot, slot + 1, distinct_sequences=True).clone()
    same = torch.equal(outs[0], outs[1])
    print(f"{weights} {rows:5d} rows: bf16 model {t16:7.3f} ms; expansion in line {min(ts[0]):7.3f} ms; on the side stream {min(ts[1]):7.3f} ms; same bits {same}", flush=True)
L.mn_wide_tune_dequant(1)
