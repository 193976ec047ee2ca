"""A/B of RMSNorm(ln1) in the decoder chain at the reference's call shape (2-4 CFG rows of one image, full 28-layer 16B-A3B stack): the
QKV streaming launch normalising the residual stream itself (stream_mfma.hip FUSE_RMSNORM) against the one-workgroup-per-row glue
launch in front of it — ms per decoder step, interleaved arms, and whether the hidden states are the same bits.
    python tools/exp/llm_ln1_fuse_ab.py [bf16|fp8|int8|int4]"""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_llm_tune_chain.argtypes = [ctypes.c_int]; L.mn_llm_tune_chain.restype = None
dev = torch.device("cuda", 0)
weights = sys.argv[1] if len(sys.argv) > 1 else "bf16"
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=1, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
if weights != "bf16":
    dec = dec.to_fp8(n_seq=4, weights=weights)
g = torch.Generator(device=dev).manual_seed(1)
small = dec.view(t_max=200, n_seq=4)
for rows in (2, 3, 4):
    x = torch.randn(rows, cfg.hidden_size, device=dev, generator=g)
    seq = torch.arange(rows, dtype=torch.int32, device=dev); slot = torch.full((rows,), 60, dtype=torch.int32, device=dev)
    def run(): return small.step(x, seq, slot, slot, slot + 1, distinct_sequences=True)
    res = {}
    for rnd in range(3):
        for arm in (32 | (1 << 17), 32):
            L.mn_llm_tune_chain(arm)
            run(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): out = run()
            e.record(); torch.cuda.synchronize()
            res.setdefault(arm, []).append((s.elapsed_time(e) / 20, out.clone()))
    a, b = res[32 | (1 << 17)], res[32]
    print(f"{weights} {rows} rows: glue launch + QKV {min(t for t, _ in a):.3f} ms ({', '.join('%.3f' % t for t, _ in a)});  QKV with the RMSNorm prologue "
          f"{min(t for t, _ in b):.3f} ms ({', '.join('%.3f' % t for t, _ in b)});  same bits {torch.equal(a[0][1], b[0][1])} (max diff {((a[0][1] - b[0][1]).abs().max() / a[0][1].abs().max()).item():.1e}), finite {bool(torch.isfinite(b[0][1]).all())}", flush=True)
L.mn_llm_tune_chain(32)
