"""A/B of the K-complete RF ResBlock chain (stream_kc.hip: w12' with the LayerNorm-modulate prologue and the SwiGLU epilogue, w3' with the
gated-residual epilogue — two launches per block, no slabs, no glue) against round 4's three-launch chain at the reference's call shape
— 1 image, 1 / 2 CFG rows — full 16B-A3B RF head, every weight format: ms per RF sampler call (16 Euler steps x 12 blocks) interleaved in
one process, and the difference of the sampled latents between the two forms (same weights, same inputs).
    python tools/exp/rf_kc_ab.py [bf16,fp8,int8,int4] [rows,...]"""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_rf_tune_fuse.argtypes = [ctypes.c_int]; L.mn_rf_tune_fuse.restype = None
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
args = argparse.Namespace(tiny=False, tokens=256, layers=2, prompt_len=40, images=1, cfg_rows=2, weights="bf16")
cfg, dec, rf0, tok = bench.build_models(args, dev, 0)
del dec, tok
WEIGHTS = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("bf16", "fp8", "int8", "int4")
ROWS = tuple(int(r) for r in sys.argv[2].split(",")) if len(sys.argv) > 2 else (2, 1, 3)
for weights in WEIGHTS:
    rf = rf0 if weights == "bf16" else rf0.to_fp8(weights)
    for rows in ROWS:
        hid = torch.randn(rows, cfg.hidden_size, device=dev, generator=g)
        noise = torch.randn(1, 32, device=dev, generator=g)
        res = {}
        for rnd in range(3):                       # interleaved rounds: box drift shows as spread between rounds
            for on in (3 | 8, 3):                  # bit 3 set = K-complete OFF
                L.mn_rf_tune_fuse(on)
                lat = torch.empty(1, 32, device=dev)
                t = ev(lambda: rf.sample(hid, noise, n_images=1, out=lat))
                res.setdefault(on, []).append((t, lat.clone()))
        old, new = res[3 | 8], res[3]
        d = (new[0][1] - old[0][1]).abs().max().item() / old[0][1].abs().max().item()
        print(f"{weights} rows {rows}: three launches per block {min(t for t, _ in old):6.3f} ms ({', '.join('%.3f' % t for t, _ in old)}), "
              f"K-complete {min(t for t, _ in new):6.3f} ms ({', '.join('%.3f' % t for t, _ in new)})  latents differ by {d:.2e} (max-norm, relative)", flush=True)
    if weights != "bf16":
        del rf
        torch.cuda.empty_cache()
L.mn_rf_tune_fuse(3)
