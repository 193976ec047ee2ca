"""One shape of the wide-row GEMM, a few launches, for rocprofv3 (--kernel-trace --stats, or --pmc passes).
usage: prof_gemm256.py {w12|w3|ada|sq|f8w12} [rows]      (f8w12: the fp8-MFMA regime's w12 launch, mingnative.h section 8)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd._lib import lib, ptr, check, current_stream
L = lib()
which = sys.argv[1] if len(sys.argv) > 1 else "w12"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = "cuda"
g = torch.Generator().manual_seed(0)
def split(x):
    hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()
if which == "w12":
    K, hid = 3072, 8192
    a2 = split(torch.randn(rows, K, generator=g).to(dev))
    ws = [(torch.randn(2 * hid, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev) for _ in range(6)]
    y = torch.empty(2, rows, hid, dtype=torch.bfloat16, device=dev)
    fn = lambda i: check(L.mn_gemm256_swiglu_split(ptr(a2), K, a2.stride(0), ptr(ws[i % 6]), K, None, ptr(y), hid, y.stride(0), rows, hid, K, current_stream()), "x")
elif which == "f8w12":
    from ming_univision_amd import ops
    K, hid = 3072, 8192
    x8, xs = ops.quant_rows(torch.randn(rows, K, generator=g).to(dev).to(torch.bfloat16), "fp8")
    w8 = [ops.quant_rows((torch.randn(2 * hid, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev), "fp8") for _ in range(6)]
    y = torch.empty(rows, hid, dtype=torch.bfloat16, device=dev)
    fn = lambda i: L.mn_gemm256_f8(ptr(x8), K, ptr(xs), ptr(w8[i % 6][0]), K, ptr(w8[i % 6][1]), None, ptr(y), hid, rows, hid, K, 1, 1, current_stream())
elif which == "w3":
    K, N = 8192, 3072
    a2 = split(torch.randn(rows, K, generator=g).to(dev))
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev) for _ in range(6)]
    ks = max(1, 256 // (((rows + 127) // 128) * 12))
    P = torch.empty(ks + 1, rows, N, dtype=torch.float32, device=dev)
    fn = lambda i: L.mn_gemm256_splitk(ptr(a2), K, a2.stride(0), ptr(ws[i % 6]), K, None, ptr(P), rows, N, K, ks, current_stream())
elif which == "ada":
    K, N, M = 3072, 116736, 16 * rows
    a2 = split(torch.randn(M, K, generator=g).to(dev))
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)]
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    fn = lambda i: check(L.mn_gemm256(ptr(a2), K, a2.stride(0), ptr(ws[0]), K, None, ptr(out), N, M, N, K, 2, current_stream()), "x")
else:
    M = N = K = rows
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fn = lambda i: check(L.mn_gemm256(ptr(a), K, 0, ptr(ws[0]), K, None, ptr(out), N, M, N, K, 0, current_stream()), "x")
n = 4 if which == "ada" else 24
for i in range(n): fn(i)
torch.cuda.synchronize()
print("done", which, rows)
