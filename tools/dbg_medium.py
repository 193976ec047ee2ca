import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import ops
for (M, N, K) in [(12, 64, 256), (12, 64, 64), (12, 176, 64), (12, 64, 176), (12, 32, 64), (12, 64, 32), (16, 64, 256)]:
    x = torch.randn(M, K, device="cuda")
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    y = ops.skinny_gemm(x, w)
    torch.cuda.synchronize()
    ref = x.double() @ w.double().T
    print(M, N, K, float((y.double() - ref).abs().max()), flush=True)
