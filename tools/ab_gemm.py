"""bf16 MFMA GEMM: global_load_lds staging vs register staging (in-process A/B) + correctness."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
from ming_univision_amd import ops
from ming_univision_amd._lib import lib
L = lib()
L.mn_gemm_tune.argtypes = [ctypes.c_int]; L.mn_gemm_tune.restype = None
def run(M, N, K, epi="bf16"):
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    ref = (a.double() @ w.double().T)
    res = {}
    for glds in (0, 1):
        L.mn_gemm_tune(glds)
        out = ops.gemm_bf16(a, w, None, epi)
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        for _ in range(3): ops.gemm_bf16(a, w, None, epi, out=out)
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm_bf16(a, w, None, epi, out=out)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 10)
        ts.sort(); res[glds] = (ts[2], err)
    f = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} {epi}: reg {res[0][0]*1e3:.1f} us {f/res[0][0]/1e9:.0f} TF (err {res[0][1]:.1e}) | glds {res[1][0]*1e3:.1f} us {f/res[1][0]/1e9:.0f} TF (err {res[1][1]:.1e})", flush=True)
for shp in [(4096, 4096, 4096), (8192, 8192, 8192), (1024, 116736, 3072), (16384, 4096, 1024), (16384, 1024, 4096), (4160, 2304, 768), (4160, 4096, 768), (1058, 3072, 2048), (100, 2816, 2048), (50, 1000, 192)]:
    run(*shp, epi="f32" if shp[0] == 1024 else "bf16")
