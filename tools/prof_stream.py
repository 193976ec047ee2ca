"""Launch the streaming MFMA kernel at the RF w12 shape (rows from argv, default 32) 24 times for PMC collection."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd._lib import lib, ptr, current_stream
M, N2, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 16384, 3072
ws = [torch.randn(N2, K, device="cuda").to(torch.bfloat16) for _ in range(6)]
Y = torch.randn(2 * M, K, device="cuda").to(torch.bfloat16)
nz = lib().mn_stream_mfma_slices(M, N2, K)
P = torch.empty(nz * M * N2, device="cuda")
for i in range(24):
    lib().mn_stream_mfma(ptr(Y), ptr(ws[i % 6]), ptr(P), M, N2, K, current_stream())
torch.cuda.synchronize()
