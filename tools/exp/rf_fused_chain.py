"""A/B of the fused RF ResBlock chain (stream_fuse.h: the SwiGLU glue folded into w3's prologue, three launches per block) against the
four-launch chain at the reference's call shape — 1 image, 2 or 3 CFG rows — full 16B-A3B RF head, bf16 and fp8 weights: ms per RF sampler call (16 Euler steps x 12 blocks),
and the difference of the sampled latents between the two forms (same weights, same inputs)."""
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
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
del dec, tok
WEIGHTS = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("bf16", "fp8")
ROWS = tuple(int(r) for r in sys.argv[2].split(",")) if len(sys.argv) > 2 else (2, 3, 4, 1, 16, 64)
for weights in WEIGHTS:
    if weights == "fp8":
        rf = rf.to_fp8()
    for rows, n_img in [(r, r // 2 if r >= 4 else 1) for r in ROWS]:
        hid = torch.randn(rows, cfg.hidden_size, device=dev, generator=g)
        noise = torch.randn(n_img, 32, device=dev, generator=g)
        res = {}
        for on in (0, 1, 3, 7):
            L.mn_rf_tune_fuse(on)
            lat = torch.empty(n_img, 32, device=dev)
            t = ev(lambda: rf.sample(hid, noise, n_images=n_img, out=lat))
            res[on] = (t, lat.clone())
        d = max((res[7][1] - res[k][1]).abs().max().item() / res[7][1].abs().max().item() for k in (0, 1, 3))
        print(f"{weights} rows {rows}: four launches per block {res[0][0]:6.3f} ms, three {res[1][0]:6.3f} ms, + one-launch step boundary {res[3][0]:6.3f} ms "
              f"(bf16 adaLN through the GEMM instead of the streaming launch: {res[7][0]:6.3f})  latents differ by {d:.2e} (max-norm, relative)", flush=True)
L.mn_rf_tune_fuse(3)
