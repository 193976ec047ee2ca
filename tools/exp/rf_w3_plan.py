"""RF sampler at 2 CFG rows under launch-plan overrides of the w3 launches only (K = 8192: slices of 256 x kch k, nw waves per workgroup):
fewer K-slices = fewer slabs for the residual glue to sum (its 196 KB per row come through one CU) against longer per-wave chains in w3."""
import os, sys, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_stream_tune_plan_k.argtypes = [ctypes.c_int] * 3; L.mn_stream_tune_plan_k.restype = None
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
for weights in ("bf16", "fp8"):
    head = rf if weights == "bf16" else rf.to_fp8()
    hid = torch.randn(2, cfg.hidden_size, device=dev, generator=g)
    noise = torch.randn(1, 32, device=dev, generator=g)
    for kch, nw in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (4, 8), (2, 16), (2, 8)):
        L.mn_stream_tune_plan_k(8192, kch, nw)
        head._ws = {}
        lat = torch.empty(1, 32, device=dev)
        try:
            t = ev(lambda: head.sample(hid, noise, n_images=1, out=lat))
            print(f"{weights} w3 plan (kch, nw) = ({kch}, {nw}): {t:6.3f} ms per sampler call", flush=True)
        except Exception as ex:
            print(f"{weights} w3 plan ({kch}, {nw}): {ex}", flush=True)
L.mn_stream_tune_plan_k(0, 0, 0)
