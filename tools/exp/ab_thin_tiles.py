"""Expert GEMMs over the tile list with / without the MFMAs of dead M-fragments (mn_gemm256_tune_thin, dev library): the lock-step token
loop at the bench's operating point, arms interleaved, plus the RF sampler alone (whose launches share the kernel instantiations)."""
import sys, os, argparse, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_gemm256_tune_thin.argtypes = [ctypes.c_int]; L.mn_gemm256_tune_thin.restype = None
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
args = argparse.Namespace(tiny=False, tokens=NT, layers=None, prompt_len=40, images=B, cfg_rows=2)
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
g = torch.Generator(device=dev).manual_seed(0)
prompt = torch.randint(0, 100000, (B, 40), generator=g, device=dev)
noises = torch.randn(B, NT + 1, 32, generator=g, device=dev)
hid = torch.randn(2 * B, cfg.hidden_size, device=dev, generator=g)
noise = torch.randn(B, rf.target, device=dev, generator=g)
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = bench.one_image(cfg, dec, rf, tok, prompt, noises, 1, 2)
    torch.cuda.synchronize(); return time.perf_counter() - t0, out["latents"].clone()
def t_rf():
    rf.sample(hid, noise, n_images=B); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): rf.sample(hid, noise, n_images=B)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3 * 1e3
run()
ref = None
for rnd in range(3):
    for on in (0, 1):
        L.mn_gemm256_tune_thin(on)
        t, lat = run()
        if ref is None: ref = lat
        print(f"round {rnd} skip dead fragments {on}: {t:.3f} s for {NT} tokens of {B} images = {B * NT / t:.0f} tokens/s; RF sampler {t_rf():.2f} ms; "
              f"latents identical to arm 0: {torch.equal(lat, ref)}", flush=True)
L.mn_gemm256_tune_thin(1)
