"""A/B at the reference's call shape (2 / 3 CFG rows of one image) of the two forms of a decoder step on the full 28-layer 16B-A3B stack:
the chain of weight-streaming MFMA launches + glue kernels (engine.hip, llm_chain_ok) against the 1-row-style sequence of fp32-FMA
launches with fused prologues / epilogues and the one-launch router: ms per step, and the difference of the hidden states."""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_llm_tune_chain.argtypes = [ctypes.c_int]; L.mn_llm_tune_chain.restype = None
L.mn_moe_tune_gate_up.argtypes = [ctypes.c_int, ctypes.c_int]; L.mn_moe_tune_gate_up.restype = None
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
        for mx in (32, 32 | (1 << 16), 1, 2):
            L.mn_llm_tune_chain(1 if mx == 2 else (32 if mx == 32 | (1 << 16) else mx))
            # arm 32: the shipped chain (router + gate/up in one launch at 2 rows); arm 32|1<<16: the chain with the separate router launch and the pair launch;
            # arm 1: the fp32-FMA sequence; arm 2: that sequence with the one-launch router + gate/up at every row count
            L.mn_moe_tune_gate_up(1, (4 if mx == 2 else 1) | ((0 if mx == 32 | (1 << 16) else 2) << 8) | (1 << 16))      # (bit 16: the one-launch form for bf16 / e4m3 too)
            run(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): out = run()
            e.record(); torch.cuda.synchronize()
            res.setdefault(mx, []).append((s.elapsed_time(e) / 20, out.clone()))
    d = (res[32][0][1] - res[1][0][1]).abs().max().item() / res[32][0][1].abs().max().item()
    old = res[32 | (1 << 16)]
    d2 = (res[32][0][1] - old[0][1]).abs().max().item() / res[32][0][1].abs().max().item()
    print(f"{weights} {rows} rows: chain with the router launch + pair launch {min(t for t, _ in old):.3f} ms ({', '.join('%.3f' % t for t, _ in old)}), differs by {d2:.1e}; "
          f"chain with router + gate/up in ONE launch {min(t for t, _ in res[32]):.3f} ms ({', '.join('%.3f' % t for t, _ in res[32])}); "
          f"fp32-FMA sequence {min(t for t, _ in res[1]):.3f} ms ({', '.join('%.3f' % t for t, _ in res[1])}); "
          f"fp32-FMA sequence with the one-launch router + gate/up {min(t for t, _ in res[2]):.3f} ms ({', '.join('%.3f' % t for t, _ in res[2])}); hidden states differ by {d:.1e}", flush=True)
L.mn_llm_tune_chain(32)
L.mn_moe_tune_gate_up(1, 1 | (2 << 8))
