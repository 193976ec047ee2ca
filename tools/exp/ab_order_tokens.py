"""End-to-end A/B of the gemm256 tile order on the lock-step token loop: 768 images, 25 visual tokens per run, arms interleaved."""
import sys, os, argparse, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_gemm256_tune_order.argtypes = [ctypes.c_int, ctypes.c_int]; L.mn_gemm256_tune_order.restype = None
dev = torch.device("cuda", 0)
B, NT = 768, 25
args = argparse.Namespace(tiny=False, tokens=NT, layers=None, prompt_len=40, images=B, cfg_rows=2)
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
assert cfg.num_image_tokens_for_gen == NT
g = torch.Generator(device=dev).manual_seed(0)
prompt = torch.randint(0, 100000, (B, 40), generator=g, device=dev)
noises = torch.randn(B, NT + 1, 32, generator=g, device=dev)
ARMS = [tuple(int(v) for v in a.split(',')) for a in (sys.argv[1:] or ['4,1', '4,4'])]
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.one_image(cfg, dec, rf, tok, prompt, noises, 1, 2)
    torch.cuda.synchronize(); return time.perf_counter() - t0
run()
for rnd in range(3):
    for gm in ARMS:
        L.mn_gemm256_tune_order(*gm)
        t = run()
        print(f"round {rnd} (dense, list) bands {gm}: {t:.3f} s for {NT} tokens of {B} images = {B * NT / t:.0f} tokens/s", flush=True)
L.mn_gemm256_tune_order(4, 1)
