"""Launch the dominant kernel (RF w12 skinny GEMM at full size, rows=2) N times for PMC collection."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import ops
M, N, K = 2, 8192, 3072
ws = [torch.randn(2 * N, K, device="cuda").to(torch.bfloat16) for _ in range(6)]   # 6 x 100 MB > MALL
b = torch.zeros(2 * N, device="cuda", dtype=torch.bfloat16)
g = torch.ones(K, device="cuda", dtype=torch.bfloat16)
x = torch.randn(M, K, device="cuda"); sh = torch.randn(M, K, device="cuda"); sc = torch.randn(M, K, device="cuda")
out = torch.empty(M, N, device="cuda")
for i in range(24):
    ops.skinny_gemm(x, ws[i % 6], b, prologue="ln_mod", epilogue="swiglu", out=out, ln_g=g, ln_b=g, eps=1e-6, pro_a=sh, pro_b=sc)
torch.cuda.synchronize()
