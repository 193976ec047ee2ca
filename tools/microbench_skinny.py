"""Micro-benchmark of the weight-streaming skinny GEMM at the RF-head / LLM decode shapes.
Prints achieved algorithmic HBM GB/s per shape (weights bytes / kernel time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import ops

def bench(M, N, K, epi="none", pro="none", iters=30, nbuf=6):
    rows = 2 * N if epi == "swiglu" else N
    ws = [torch.randn(rows, K, device="cuda").to(torch.bfloat16) for _ in range(nbuf)]   # rotate > L2/MALL
    x = torch.randn(M, K, device="cuda")
    kw = {}
    if pro == "ln_mod":
        kw = dict(prologue="ln_mod", eps=1e-6, pro_a=torch.randn(M, K, device="cuda"), pro_b=torch.randn(M, K, device="cuda"))
    if epi == "resid_gate":
        kw.update(res=torch.randn(M, N, device="cuda"), gate=torch.randn(M, N, device="cuda"))
    out = torch.empty(M, N, device="cuda")
    for i in range(5):
        ops.skinny_gemm(x, ws[i % nbuf], epilogue=epi, out=out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        ops.skinny_gemm(x, ws[i % nbuf], epilogue=epi, out=out, **kw)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    gb = rows * K * 2 / 1e9
    print(f"M={M} N={N:6d} K={K:5d} epi={epi:10s} pro={pro:7s}: {us:8.1f} us  {gb/us*1e6:7.0f} GB/s  ({rows*K*2/2**20:.0f} MiB)", flush=True)

if __name__ == "__main__":
    for M in (1, 2, 3):
        bench(M, 8192, 3072, "swiglu", "ln_mod")      # RF w12
        bench(M, 3072, 8192, "resid_gate")            # RF w3
        bench(M, 12 * 9216 + 6144, 3072)              # RF adaLN (all blocks)
        bench(M, 3072, 2048)                          # LLM qkv
        bench(M, 2048, 2048)                          # LLM dense
        bench(M, 126464, 2048)                        # lm_head
    bench(8, 8192, 3072, "swiglu", "ln_mod")
