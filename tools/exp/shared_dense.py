"""would the two shared pseudo-experts run faster as ONE dense hi/lo gemm256 problem (so that the routed experts could
take the 192-row tall tile measured in round 3: gate/up 317 -> 221 us, down 160 -> 116 us on the routed experts alone)?  Times the
shared expert of a 16B-A3B layer (intermediate 2 x 1408 = 2816) at the bench's 1536 rows as dense launches: gate/up with the SwiGLU +
split epilogue, down as split-K slabs; in-process, HIP events, 5 rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
H, SI = 2048, 2816
g = torch.Generator().manual_seed(0)
x2 = ops.split_hilo(torch.randn(rows, H, generator=g).cuda())
wgu = [(torch.randn(2 * SI, H, generator=g) * H ** -0.5).to(torch.bfloat16).cuda() for _ in range(3)]
wdn = [(torch.randn(H, SI, generator=g) * SI ** -0.5).to(torch.bfloat16).cuda() for _ in range(3)]
h2 = ops.gemm256_swiglu_split(x2, wgu[0])


def timed(fn, n=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


res = {}
for r in range(5):
    res.setdefault("gate/up SwiGLU-split", []).append(timed(lambda i: ops.gemm256_swiglu_split(x2, wgu[i % 3])))
    for ks in (1, 2, 3):
        res.setdefault("down split-K %d" % ks, []).append(timed(lambda i: ops.gemm256_splitk(h2, wdn[i % 3], None, ks)))
for k, v in res.items():
    print("shared expert dense, %d rows: %s: %.1f us" % (rows, k, sorted(v)[2]), flush=True)
gu = sorted(res["gate/up SwiGLU-split"])[2]
dn = min(sorted(v)[2] for k, v in res.items() if k.startswith("down"))
print("dense shared expert: %.0f + %.0f us; + routed experts on the round-3 tall tile (221 + 116 us) = %.0f us per layer against 274 + 164 = 438 us "
      "for the shipped single grouped launches (profiles/r03_bench_default_site_stats.txt)" % (gu, dn, gu + dn + 337))
