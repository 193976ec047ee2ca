"""Band height of the gemm256 tile order (mn_gemm256_tune_order) end to end: RF sampler at 1536 rows, MingTok enc->dec 64 x 256^2."""
import sys, os, argparse, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_gemm256_tune_order.argtypes = [ctypes.c_int, ctypes.c_int]; L.mn_gemm256_tune_order.restype = None
dev = torch.device("cuda", 0)
B = 768
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=B, cfg_rows=2)
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
g = torch.Generator(device=dev).manual_seed(0)
hid = torch.randn(2 * B, cfg.hidden_size, device=dev, generator=g)
noise = torch.randn(B, 32, device=dev, generator=g)
lat = torch.empty(B, 32, device=dev)
imgs = (torch.rand(64, 3, 256, 256, device=dev, generator=g) * 2 - 1)
def wall(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(2):
    for gm in (0, 3, 4, 6, 8, 12):
        L.mn_gemm256_tune_order(gm, 1)
        t_rf = wall(lambda: rf.sample(hid, noise, n_images=B, out=lat), 2)
        t_c2 = wall(lambda: tok.forward_enc_dec(imgs), 3)
        print(f"group_m {gm:2d}: RF sample 1536 rows {t_rf:7.2f} ms   C2 {t_c2:6.2f} ms", flush=True)
L.mn_gemm256_tune_order(4, 1)
