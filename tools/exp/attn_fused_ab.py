"""A/B of RoPE + KV append fused into the split decode-attention launch (mn_llm_step_ex, MN_STEP_DISTINCT_SEQUENCES) against the separate
append launch: ms per full-depth decoder step (28 layers, 16B-A3B) at the reference's call shapes — 1 row (text decode), 2 / 3 rows (the
CFG rows of one image) — and at 16 / 64 rows, over the cache lengths an image's tokens see."""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_attn_tune_fuse.argtypes = [ctypes.c_int]; L.mn_attn_tune_fuse.restype = None
L.mn_attn_tune_one.argtypes = [ctypes.c_int]; L.mn_attn_tune_one.restype = None
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=32, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
del rf, tok
for R in (1, 2, 3, 4, 8, 16, 32, 64):
    x = torch.randn(R, cfg.hidden_size, device=dev, generator=g)
    seq = torch.arange(R, dtype=torch.int32, device=dev)
    out = torch.empty(R, cfg.hidden_size, device=dev)
    km = torch.ones(R, dec.t_max, dtype=torch.uint8, device=dev)
    for T in (40, 296):
        slot = torch.full((R,), T, dtype=torch.int32, device=dev)
        res = {}
        for one in (0, 64):
            for fuse in (0, 1):
                L.mn_attn_tune_one(one); L.mn_attn_tune_fuse(fuse)
                res[(one, fuse)] = ev(lambda: dec.step(x, seq, slot, slot, slot + 1, km, None, out=out, rows=R, distinct_sequences=True))
        print(f"rows {R:2d} cache {T:3d}: split + combine {res[(0, 0)]:6.3f} ms (append fused where allowed {res[(0, 1)]:6.3f}) | one launch {res[(64, 0)]:6.3f} "
              f"(append fused {res[(64, 1)]:6.3f})", flush=True)
L.mn_attn_tune_fuse(1); L.mn_attn_tune_one(16)
