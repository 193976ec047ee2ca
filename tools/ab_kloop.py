"""K-loop streaming kernel vs the K-slice kernel: correctness against float64 and in-process timing."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
from ming_univision_amd._lib import lib, ptr, current_stream
L = lib()
i, p = ctypes.c_int, ctypes.c_void_p
L.mn_stream_kloop.argtypes = [p, p, p, i, i, i, p]; L.mn_stream_kloop.restype = i
L.mn_stream_kloop_slices.argtypes = [i, i, i]; L.mn_stream_kloop_slices.restype = i
L.mn_stream_kloop_tune.argtypes = [i, i, i]; L.mn_stream_kloop_tune.restype = None

def check(M, N, K):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g)
    hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
    Y = torch.cat([hi, lo], 0).contiguous().cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16)
    wd = w.cuda()
    nz = L.mn_stream_kloop_slices(M, N, K)
    P = torch.full((nz, M, N), float("nan"), device="cuda")
    rc = L.mn_stream_kloop(ptr(Y), ptr(wd), ptr(P), M, N, K, current_stream())
    torch.cuda.synchronize()
    ref = (hi.double() + lo.double()) @ w.double().T
    out = P.double().sum(0).cpu()
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"check M={M} N={N} K={K}: nz={rc} rel err {err:.2e}", flush=True)
    assert err < 1e-5

def bench(M, N2, K, variants, rounds=5, iters=24):
    ws = [torch.randn(N2, K, device="cuda").to(torch.bfloat16) for _ in range(6)]
    Y = torch.randn(2 * M, K, device="cuda").to(torch.bfloat16)
    P = torch.empty(64 * M * N2, device="cuda")
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            if v[0] == "slice":
                fn = lambda w: L.mn_stream_mfma(ptr(Y), ptr(w), ptr(P), M, N2, K, current_stream())
            else:
                L.mn_stream_kloop_tune(v[1], v[2], v[3])
                fn = lambda w: L.mn_stream_kloop(ptr(Y), ptr(w), ptr(P), M, N2, K, current_stream())
            for k in range(3): fn(ws[k % 6])
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for k in range(iters): fn(ws[k % 6])
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) * 1e3 / iters)
    for v in variants:
        t = sorted(res[v]); print(f"M={M} N={N2} K={K} {v}: median {t[len(t)//2]:.1f} us ({N2*K*2/t[len(t)//2]/1e3:.0f} GB/s)", flush=True)

L.mn_stream_kloop_tune(0, 0, 0)
M = 64
bench(M, 16384, 3072, [("kloop", 0, d, 2) for d in (2, 4)] + [("kloop", 0, d, 1) for d in (2, 4)])
bench(M, 3072, 8192, [("kloop", 0, d, 2) for d in (2, 4)] + [("kloop", 0, d, 1) for d in (2, 4)])
bench(M, 3072, 2048, [("kloop", 0, d, 2) for d in (2, 4)])
