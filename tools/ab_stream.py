"""In-process interleaved A/B of the streaming MFMA kernel variants (HIP events, many rounds)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
from ming_univision_amd._lib import lib, ptr, current_stream
L = lib()
L.mn_stream_tune_plan.argtypes = [ctypes.c_int] * 2; L.mn_stream_tune_plan.restype = None
M = 16
def bench(N2, K, variants, rounds=6, iters=24):
    ws = [torch.randn(N2, K, device="cuda").to(torch.bfloat16) for _ in range(6)]
    Y = torch.randn(2 * M, K, device="cuda").to(torch.bfloat16)
    P = torch.empty(64 * M * N2, device="cuda")
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            L.mn_stream_tune_plan(v[1], v[2])
            for i in range(3): L.mn_stream_mfma(ptr(Y), ptr(ws[i % 6]), ptr(P), M, N2, K, current_stream())
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(iters): L.mn_stream_mfma(ptr(Y), ptr(ws[i % 6]), ptr(P), M, N2, K, current_stream())
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) * 1e3 / iters)
    for v in variants:
        t = sorted(res[v]); print(f"N={N2} K={K} M={M} depth={v[0]} kch={v[1]} nw={v[2]}: median {t[len(t)//2]:.1f} us  min {t[0]:.1f}  ({N2*K*2/t[len(t)//2]/1e3:.0f} GB/s)", flush=True)
V = [(1, 0, 0), (2, 0, 0), (1, 3, 8), (2, 3, 8), (1, 2, 8), (2, 2, 8)]   # ring depth, kch, nw
for M in (16, 32):
    bench(16384, 3072, V)
    bench(3072, 8192, V)
