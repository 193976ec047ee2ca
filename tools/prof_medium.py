import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def run(N, K, epi, pro):
    rows = 2 * N if epi == "swiglu" else N
    ws = [(torch.randn(rows, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(4)]
    x = torch.randn(M, K, device="cuda")
    kw = {}
    if pro == "ln_mod":
        kw = dict(prologue="ln_mod", eps=1e-6, pro_a=torch.randn(M, K, device="cuda"), pro_b=torch.randn(M, K, device="cuda"))
    if pro == "rmsnorm":
        kw = dict(prologue="rmsnorm", eps=1e-6, ln_g=torch.ones(K, device="cuda", dtype=torch.bfloat16))
    if epi == "resid_gate":
        kw.update(res=torch.randn(M, N, device="cuda"), gate=torch.randn(M, N, device="cuda"))
    for i in range(8):
        ops.skinny_gemm(x, ws[i % 4], epilogue=epi, **kw)
run(2048, 2048, "none", "none")
run(3072, 2048, "none", "rmsnorm")
run(3072, 8192, "resid_gate", "none")
run(8192, 3072, "swiglu", "ln_mod")
torch.cuda.synchronize()
