"""Grouped expert GEMM at 64 rows (64 routed experts with a handful of rows each + 2 shared experts with all rows):
K-slice kernel for <= 32 rows + K-loop for the rest (product path) vs K-loop launches by row class."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd._lib import lib, ptr, current_stream
L = lib()
i, p, i64 = ctypes.c_int, ctypes.c_void_p, ctypes.c_int64
L.mn_stream_kloop_grouped.argtypes = [p, i, p, i64, p, i, p, p, i, i, i, i, i, i, p]; L.mn_stream_kloop_grouped.restype = i
E, S, top = 64, 2, 6
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
def make(N, K, gather):
    idx = torch.stack([torch.randperm(E, generator=g)[:top] for _ in range(M)])       # [M, top]
    cnt = torch.bincount(idx.flatten(), minlength=E).tolist() + [M] * S
    off = torch.tensor([0] + list(torch.tensor(cnt).cumsum(0)), dtype=torch.int32)
    total = int(off[-1])
    xrows = torch.randint(0, M, (total,), generator=g, dtype=torch.int32)
    nx = M if gather else total
    Y = (torch.randn(2 * nx, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    W = [(torch.randn(E + S, N, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(3)]
    P = torch.empty(8 * total * N, device="cuda")
    return off.cuda(), (xrows.cuda() if gather else None), Y, W, P, total, nx, max(cnt[:E])
def bench(name, N, K, gather):
    off, xr, Y, W, P, total, nx, mx = make(N, K, gather)
    nzs = L.mn_stream_mfma_grouped_slices(E + S, M, N, K)
    def prod(w): L.mn_stream_mfma_grouped(ptr(Y), nx, ptr(w), N * K, ptr(P), total, ptr(off), ptr(xr), E + S, M, N, K, current_stream())
    def kl_all(w): L.mn_stream_kloop_grouped(ptr(Y), nx, ptr(w), N * K, ptr(P), total, ptr(off), ptr(xr), E + S, M, 0, nzs, N, K, current_stream())
    def kl_cls(w):
        L.mn_stream_kloop_grouped(ptr(Y), nx, ptr(w), N * K, ptr(P), total, ptr(off), ptr(xr), E + S, 16, 0, nzs, N, K, current_stream())
        L.mn_stream_kloop_grouped(ptr(Y), nx, ptr(w), N * K, ptr(P), total, ptr(off), ptr(xr), E + S, 32, 16, nzs, N, K, current_stream())
        if M > 32: L.mn_stream_kloop_grouped(ptr(Y), nx, ptr(w), N * K, ptr(P), total, ptr(off), ptr(xr), E + S, 64, 32, nzs, N, K, current_stream())
    def kl_nz(n):
        return lambda w: L.mn_stream_kloop_grouped(ptr(Y), nx, ptr(w), N * K, ptr(P), total, ptr(off), ptr(xr), E + S, M, 0, n, N, K, current_stream())
    res = {}
    for r in range(5):
        for nm, fn in (("slice+kloop", prod), ("kloop nz1", kl_nz(1)), ("kloop nz2", kl_nz(2)), ("kloop nz3", kl_nz(3)), ("kloop nz4", kl_nz(4))):
            for k in range(2): fn(W[k % 3])
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for k in range(9): fn(W[k % 3])
            e.record(); torch.cuda.synchronize()
            res.setdefault(nm, []).append(s.elapsed_time(e) * 1e3 / 9)
    gb = (E + S) * N * K * 2 / 1e9
    print(name, f"max routed rows {mx}, slabs {nzs}:", "  ".join(f"{k}: {sorted(v)[2]:.0f} us ({gb / sorted(v)[2] * 1e3:.2f} TB/s)" for k, v in res.items()), flush=True)
bench("gate_up", 2816, 2048, True)
bench("down", 2048, 1408, False)
