"""Does a second pass over the same 100 MB weight matrix (served by the 256 MiB Infinity Cache) run faster than a first
pass from HBM?  Streaming MFMA kernel, RF w12 shape, 32 rows; pattern A = 12 distinct matrices in turn, pattern B =
each matrix twice in a row."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd._lib import lib, ptr, current_stream
L = lib()
M, N2, K = 32, 16384, 3072
ws = [torch.randn(N2, K, device="cuda").to(torch.bfloat16) for _ in range(12)]
Y = torch.randn(2 * M, K, device="cuda").to(torch.bfloat16)
P = torch.empty(L.mn_stream_mfma_slices(M, N2, K) * M * N2, device="cuda")
def run(order, rounds=7):
    res = []
    for r in range(rounds):
        for i in order[:6]: L.mn_stream_mfma(ptr(Y), ptr(ws[i]), ptr(P), M, N2, K, current_stream())
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in order: L.mn_stream_mfma(ptr(Y), ptr(ws[i]), ptr(P), M, N2, K, current_stream())
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) * 1e3 / len(order))
    res.sort(); return res[len(res) // 2]
a = run(list(range(12)) * 2)
b = run([i for i in range(12) for _ in range(2)])
c = run([0] * 24)
print(f"distinct matrices in turn: {a:.1f} us/launch; each twice in a row: {b:.1f} us/launch (second pass = {2*b-a:.1f} us); same matrix always: {c:.1f} us")
