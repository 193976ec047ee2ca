"""Ring-depth sweep of the K-complete RF launches (dev library): ms per RF sampler call at 2 rows."""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_rf_tune_fuse.argtypes = [ctypes.c_int]; L.mn_rf_tune_fuse.restype = None
L.mn_rf_kc_tune.argtypes = [ctypes.c_int, ctypes.c_int]; L.mn_rf_kc_tune.restype = None
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
for weights in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16", "fp8"]):
    rf = rf0 if weights == "bf16" else rf0.to_fp8(weights)
    hid = torch.randn(2, cfg.hidden_size, device=dev, generator=g)
    noise = torch.randn(1, 32, device=dev, generator=g)
    lat = torch.empty(1, 32, device=dev)
    L.mn_rf_tune_fuse(3 | 8)
    base = min(ev(lambda: rf.sample(hid, noise, n_images=1, out=lat)) for _ in range(3))
    L.mn_rf_tune_fuse(3)
    out = []
    for rd12 in (1, 2, 3):
        for rd3 in (0, 1, 2, 4):                  # 0: the default by format
            L.mn_rf_kc_tune(rd12, rd3)
            out.append((min(ev(lambda: rf.sample(hid, noise, n_images=1, out=lat)) for _ in range(2)), rd12, rd3))
    print(weights, "three-launch chain %.3f ms | K-complete (ms, rd12, rd3):" % base, " ".join("%.3f/%x/%d" % o for o in out), flush=True)
L.mn_rf_kc_tune(1, 0)
