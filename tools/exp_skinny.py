import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import ops
from ming_univision_amd._lib import lib
L = lib(); L.mn_skinny_tune.argtypes = [ctypes.c_int] * 3; L.mn_skinny_tune.restype = None

def t(M, N, K, epi, pro, iters=48, nbuf=6):
    rows = 2 * N if epi == "swiglu" else N
    ws = [torch.randn(rows, K, device="cuda").to(torch.bfloat16) for _ in range(nbuf)]
    x = torch.randn(M, K, device="cuda")
    kw = {}
    if pro == "ln_mod":
        kw = dict(prologue="ln_mod", eps=1e-6, pro_a=torch.randn(M, K, device="cuda"), pro_b=torch.randn(M, K, device="cuda"))
    if epi == "resid_gate":
        kw.update(res=torch.randn(M, N, device="cuda"), gate=torch.randn(M, N, device="cuda"))
    out = torch.empty(M, N, device="cuda")
    for i in range(4): ops.skinny_gemm(x, ws[i % nbuf], epilogue=epi, out=out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): ops.skinny_gemm(x, ws[i % nbuf], epilogue=epi, out=out, **kw)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"M={M} N={N} K={K} {epi:10s} {pro:7s}: {us:6.1f} us  {rows*K*2/us/1e3:6.0f} GB/s", flush=True)

for M in (1, 2, 4):
    t(M, 8192, 3072, "swiglu", "ln_mod")
    t(M, 8192, 3072, "swiglu", "none")
    t(M, 16384, 3072, "none", "none")
    t(M, 16384, 3072, "none", "ln_mod")
    t(M, 3072, 8192, "resid_gate", "none")
    t(M, 3072, 8192, "none", "none")
    t(M, 4096, 8192, "none", "none")
    t(M, 6144, 8192, "none", "none")
    t(M, 32768, 3072, "none", "none")
    t(M, 65536, 3072, "none", "none")
