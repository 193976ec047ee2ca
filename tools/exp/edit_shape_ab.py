"""One image at the editing call shape (3 CFG rows) and at 2 rows, full 16B-A3B stack (bench.py's batch-1 legs): visual tokens/s of REAL
image generation (CFG rows of one image select overlapping experts, unlike random rows) with the decoder's expert route switched at 3 rows
and with the (row, expert) pair launches at every row count, interleaved.
    python tools/exp/edit_shape_ab.py [bf16|fp8]"""
import sys, os, argparse, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_moe_tune_min_rows.argtypes = [ctypes.c_int]; L.mn_moe_tune_min_rows.restype = None
dev = torch.device("cuda", 0)
weights = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ARMS = tuple(int(a) for a in sys.argv[2].split(",")) if len(sys.argv) > 2 else (3, 64)      # first row count on the grouped route, per arm
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=2, cfg_rows=3, weights=weights)
L.mn_moe_tune_min_rows(3)
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
g = torch.Generator(device=dev).manual_seed(1)
prompt = torch.randint(0, 100000, (2, 40), device=dev, generator=g)
noises = torch.randn(2, 257, 32, device=dev, generator=g)
for images, rows in ((1, 3), (2, 2), (1, 2)):
    res = {}
    for rnd in range(3):
        for arm in ARMS:
            L.mn_moe_tune_min_rows(arm)
            bench.one_image(cfg, dec, rf, tok, prompt[:images], noises[:images], 1, rows); torch.cuda.synchronize()
            t0 = time.perf_counter()
            bench.one_image(cfg, dec, rf, tok, prompt[:images], noises[:images], 1, rows); torch.cuda.synchronize()
            res.setdefault(arm, []).append(256 * images / (time.perf_counter() - t0))
    print(f"{weights} {images} image(s) x {rows} CFG rows = {images * rows} rows per step: grouped route from {ARMS[0]} rows {max(res[ARMS[0]]):.2f} tok/s ({', '.join('%.2f' % t for t in res[ARMS[0]])});  grouped route from {ARMS[1]} rows {max(res[ARMS[1]]):.2f} tok/s ({', '.join('%.2f' % t for t in res[ARMS[1]])})", flush=True)
L.mn_moe_tune_min_rows(0)
