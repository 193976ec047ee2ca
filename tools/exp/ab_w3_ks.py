"""Split-K request of the few-tile hi/lo GEMMs of the wide route at the bench's row count (RF w3: N = 3072, K = 8192; QKV; dense):
time per launch by HIP events for ks = 1..8 (+ the slab reduce the consumer pays: ks * rows * N * 4 bytes read).  The shipped
choice comes from rf_wide_ksplit's cost model (wide_rf.inl)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in (("RF w3", 3072, 8192), ("LLM qkv", 3072, 2048), ("LLM dense", 2048, 2048), ("semdec w3", 1024, 2752 + 64 - 2752 % 64)):
    a2 = (torch.randn(2, rows, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    a2[1] *= 2.0 ** -9
    ws = [(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16) for _ in range(6)]
    line = []
    for ks in range(1, 9):
        if K // 64 // ks < 4:
            break
        for w in ws:
            P = ops.gemm256_splitk(a2, w, None, ks)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(48):
            P = ops.gemm256_splitk(a2, ws[i % 6], None, ks)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 48
        line.append("ks=%d(nz=%d) %.1f us %.0f TF/s" % (ks, P.shape[0], us, 2.0 * rows * N * K / us * 1e-6))
    print("%-10s rows=%d: %s" % (name, rows, " | ".join(line)), flush=True)
