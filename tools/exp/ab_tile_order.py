"""gemm256 tile order A/B (mn_gemm256_tune_order): RF w12 (SwiGLU-split), w3 (split-K 3) and adaLN shapes at 1536 rows."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
from ming_univision_amd import ops
from ming_univision_amd._lib import lib
L = lib()
L.mn_gemm256_tune_order.argtypes = [ctypes.c_int, ctypes.c_int]; L.mn_gemm256_tune_order.restype = None
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s): return (torch.randn(*s, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
a_w = rnd(2, rows, 3072); a_h = rnd(2, rows, 8192); a_ada = rnd(2, 4 * rows, 3072)
w12 = [rnd(16384, 3072) for _ in range(4)]; w3 = [rnd(3072, 8192) for _ in range(4)]; wada = rnd(16384, 3072)
def ev(fn, n=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
cases = {"w12 swiglu-split": lambda i=0: ops.gemm256_swiglu_split(a_w, w12[i % 4]),
         "w3 split-K 3": lambda i=0: ops.gemm256_splitk(a_h, w3[i % 4], None, 3),
         "adaLN-like [4 rows x 16384]": lambda i=0: ops.gemm256(a_ada, wada, None, "f32")}
for name, f in cases.items():
    res = []
    for rnd_ in range(5):
        for gm in (0, 2, 3, 4, 5, 6):
            L.mn_gemm256_tune_order(gm, 1)
            res.append((gm, ev(f)))
    L.mn_gemm256_tune_order(4, 1)
    by = {}
    for gm, t in res: by.setdefault(gm, []).append(t)
    print(name, {gm: "%.1f us" % sorted(v)[len(v) // 2] for gm, v in by.items()}, flush=True)
