"""Sweep the skinny-GEMM launch plan (R rows/wave, threads/block, blocks/CU) on the hot shapes."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
from ming_univision_amd import ops
from ming_univision_amd._lib import lib
L = lib()
L.mn_skinny_tune.argtypes = [ctypes.c_int] * 3
L.mn_skinny_tune.restype = None

def run(M, N, K, epi, pro, iters=24, nbuf=6):
    rows = 2 * N if epi == "swiglu" else N
    ws = [torch.randn(rows, K, device="cuda").to(torch.bfloat16) for _ in range(nbuf)]
    x = torch.randn(M, K, device="cuda")
    kw = {}
    if pro == "ln_mod":
        kw = dict(prologue="ln_mod", eps=1e-6, pro_a=torch.randn(M, K, device="cuda"), pro_b=torch.randn(M, K, device="cuda"))
    elif pro == "rmsnorm":
        kw = dict(prologue="rmsnorm", eps=1e-6, ln_g=torch.ones(K, device="cuda", dtype=torch.bfloat16))
    if epi == "resid_gate":
        kw.update(res=torch.randn(M, N, device="cuda"), gate=torch.randn(M, N, device="cuda"))
    out = torch.empty(M, N, device="cuda")
    def t():
        for i in range(4): ops.skinny_gemm(x, ws[i % nbuf], epilogue=epi, out=out, **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters): ops.skinny_gemm(x, ws[i % nbuf], epilogue=epi, out=out, **kw)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / iters
    res = []
    for nt in (256, 512, 768):
        for bpc in (1, 2, 4, 8):
            if nt * bpc > 2048: continue
            for R in (1, 2, 4):
                if epi == "swiglu" and R == 4: continue
                L.mn_skinny_tune(R, nt, bpc)
                try:
                    res.append((t(), nt, bpc, R))
                except Exception as ex:
                    pass
    L.mn_skinny_tune(0, 0, 0)
    base = t()
    res.sort()
    gb = rows * K * 2 / 1e9
    print(f"M={M} N={N} K={K} {epi}/{pro}: heuristic {base:.1f}us ({gb/base*1e6:.0f} GB/s); best: " +
          ", ".join(f"{u:.1f}us nt={nt} bpc={b} R={R}" for u, nt, b, R in res[:5]), flush=True)

for M in (1, 2, 3):
    run(M, 8192, 3072, "swiglu", "ln_mod")
    run(M, 3072, 8192, "resid_gate", "none")
    run(M, 2736, 1024, "swiglu", "none")
    run(M, 1024, 2736, "none", "none")
