"""Shares of one bench step (768 images) outside the lock-step token loop: prompt prefill and pixel decode."""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=B, cfg_rows=2)
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
g = torch.Generator(device=dev).manual_seed(0)
prompt = torch.randint(0, 100000, (B, 40), generator=g, device=dev)
def wall(fn, n=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
t_pre = wall(lambda: dec.prefill_many(dec.embed(prompt).reshape(B, 40, -1), [2 * i for i in range(B)]))
sem = torch.randn(B, 256, tok.feature_dim, device=dev, generator=g)
t_pix = wall(lambda: tok.forward_pixel_decoder(sem), 1)
print(f"{B} images: prompt prefill (40 tokens each, lock-step) {t_pre * 1e3:.0f} ms, pixel decode {t_pix * 1e3:.0f} ms "
      f"({723.2e9 * B / t_pix / 1e12:.0f} TFLOP/s)", flush=True)
