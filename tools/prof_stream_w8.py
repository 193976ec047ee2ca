"""Launch the fp8 streaming MFMA kernel at the RF w12 shape (rows from argv, default 2) 24 times for PMC collection / rocprofv3 --stats;
`bf16` as second argument launches the bf16 form on the same shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import ops
from ming_univision_amd._lib import lib, ptr, current_stream
M, N2, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 2), 16384, 3072
fp8 = not (len(sys.argv) > 2 and sys.argv[2] == "bf16")
ws = [(torch.randn(N2, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(6)]
qs = [ops.quant_fp8_rows(w) for w in ws]
Y = (torch.randn(2 * M, K, device="cuda") * 0.5).to(torch.bfloat16)
nz = lib().mn_stream_mfma_w8_slices(M, N2, K) if fp8 else lib().mn_stream_mfma_slices(M, N2, K)
P = torch.empty(nz * M * N2, device="cuda")
for i in range(24):
    if fp8:
        lib().mn_stream_mfma_w8(ptr(Y), ptr(qs[i % 6][0]), ptr(qs[i % 6][1]), ptr(P), M, N2, K, current_stream())
    else:
        lib().mn_stream_mfma(ptr(Y), ptr(ws[i % 6]), ptr(P), M, N2, K, current_stream())
torch.cuda.synchronize()
print("done", M, "fp8" if fp8 else "bf16", "slices", nz)
