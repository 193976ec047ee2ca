"""A/B of the persistent per-Euler-step launch (stream_kc.hip: the 12 ResBlocks = 24 phases of a step in ONE launch, grid barriers
between the phases, the next phase's first weight chunks requested before the wait) against the same phases as 24 launches, at the
reference's call shape — 1 image, 1 / 2 CFG rows — full 16B-A3B RF head, every weight format: ms per RF sampler call (16 Euler steps)
interleaved in one process, and whether the sampled latents are the same BITS (the per-step launch: same arithmetic in the same order,
they must be; the whole-sampler launch computes the final layer in fp32 FMAs instead of hi/lo MFMA slabs: equal to ~1e-6) — and the whole sampler as one launch.
    python tools/exp/rf_persist_ab.py [bf16,fp8,int8,int4] [rows,...]"""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_rf_tune_fuse.argtypes = [ctypes.c_int]; L.mn_rf_tune_fuse.restype = None
L.mn_rf_kc_persist_all.argtypes = [ctypes.c_int]; L.mn_rf_kc_persist_all.restype = None
L.mn_rf_kc_persist_all(1)                          # (A/B every format; the product enables bf16 and e4m3)
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
ROWS = tuple(int(r) for r in sys.argv[2].split(",")) if len(sys.argv) > 2 else (2, 1)
for weights in WEIGHTS:
    rf = rf0 if weights == "bf16" else rf0.to_fp8(weights)
    for rows in ROWS:
        hid = torch.randn(rows, cfg.hidden_size, device=dev, generator=g)
        noise = torch.randn(1, 32, device=dev, generator=g)
        res = {}
        ARMS = (("three-launch chain per ResBlock (K-complete off)", 3 | 8), ("24 launches per step", 3 | 16), ("one persistent launch per step", 3 | 32), ("the whole sampler in one launch", 3))
        for rnd in range(3):                       # interleaved rounds: box drift shows as spread between rounds
            for name, on in ARMS:                  # bit 4 set = persistent launch OFF, bit 5 = whole-sampler launch OFF
                L.mn_rf_tune_fuse(on)
                lat = torch.empty(1, 32, device=dev)
                t = ev(lambda: rf.sample(hid, noise, n_images=1, out=lat))
                res.setdefault(name, []).append((t, lat.clone()))
        base = res[ARMS[1][0]]
        line = f"{weights} rows {rows}:"
        for name, _ in ARMS:
            r = res[name]
            same = all(torch.equal(a[1], b[1]) for a in base for b in r)
            d = max(((a[1] - b[1]).abs().max() / a[1].abs().max()).item() for a in base for b in r)
            line += f"  {name} {min(t for t, _ in r):6.3f} ms ({', '.join('%.3f' % t for t, _ in r)})" + ("" if name == ARMS[1][0] else f" [same bits as the launches: {same}; max diff {d:.1e}; finite {bool(torch.isfinite(r[0][1]).all())}]")
        print(line, flush=True)
    if weights != "bf16":
        del rf
        torch.cuda.empty_cache()
L.mn_rf_tune_fuse(3)
