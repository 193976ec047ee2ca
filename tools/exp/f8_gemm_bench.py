"""The fp8-MFMA regime's GEMM (gemm256.hip, template F8; mingnative.h section 8) against the hi/lo bf16 pair it would replace, on the RF
head's three wide-route shapes at the bench's row count (1536) and on a square problem: HIP events over back-to-back launches, the
weights cycled so that HBM / L2 see the real footprint.  TFLOP/s are ALGORITHMIC (2 M N K); peaks: bf16 2.5 PF, fp8 5 PF dense.
    python tools/exp/f8_gemm_bench.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import ops
from ming_univision_amd._lib import lib, ptr, current_stream, check
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def ev(fn, n=40):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
L = lib()
def case(name, M, N, K, swiglu, ksplit, nw=6):
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    ws = [(torch.randn((2 if swiglu else 1) * N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
    x8, xs = ops.quant_rows(x, "fp8")
    w8 = [ops.quant_rows(w, "fp8") for w in ws]
    a2 = ops.split_hilo(x.float())
    flops = 2.0 * M * N * (2 if swiglu else 1) * K
    if swiglu:
        y8 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        y2 = torch.empty(2, M, N, dtype=torch.bfloat16, device=dev)
        f8 = lambda i: L.mn_gemm256_f8(ptr(x8), K, ptr(xs), ptr(w8[i % nw][0]), K, ptr(w8[i % nw][1]), None, ptr(y8), N, M, N, K, 1, 1, current_stream())
        hl = lambda i: check(L.mn_gemm256_swiglu_split(ptr(a2), K, a2.stride(0), ptr(ws[i % nw]), K, None, ptr(y2), N, y2.stride(0), M, N, K, current_stream()), "hl")
    else:
        nz8 = L.mn_gemm256_f8_slices(K, ksplit)
        c8 = torch.empty(nz8, M, N, dtype=torch.float32, device=dev)
        c2 = torch.empty(max(1, ksplit) + 1, M, N, dtype=torch.float32, device=dev)
        f8 = lambda i: L.mn_gemm256_f8(ptr(x8), K, ptr(xs), ptr(w8[i % nw][0]), K, ptr(w8[i % nw][1]), None, ptr(c8), N, M, N, K, 0, ksplit, current_stream())
        hl = lambda i: L.mn_gemm256_splitk(ptr(a2), K, a2.stride(0), ptr(ws[i % nw]), K, None, ptr(c2), M, N, K, ksplit, current_stream())
    t8, t2 = ev(f8), ev(hl)
    tq = ev(lambda i: ops.quant_rows(x, "fp8"))
    print(f"{name}: M={M} N={N}{'x2' if swiglu else ''} K={K} ks={ksplit} | fp8 MFMA {t8:7.1f} us = {flops / t8 * 1e-6:6.0f} TFLOP/s ({flops / t8 * 1e-6 / 5000:.3f} of the fp8 peak)"
          f" | bf16 hi/lo {t2:7.1f} us = {flops / t2 * 1e-6:6.0f} TFLOP/s ({flops / t2 * 1e-6 / 2500:.3f} of the bf16 peak) | x {t2 / t8:.2f} | + activation quantise pass {tq:.1f} us", flush=True)
case("RF w12 (SwiGLU epilogue)", rows, 8192, 3072, True, 1)
case("RF w3 (split-K slabs)", rows, 3072, 8192, False, 3)
case("RF adaLN, all Euler steps", 16 * rows, 12 * 3 * 3072 + 2 * 3072, 3072, False, 1, nw=1)
case("square", 4096, 4096, 4096, False, 1, nw=2)
