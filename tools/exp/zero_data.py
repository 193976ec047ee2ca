"""Power-limit check: the same gemm256 launch on random vs zero-filled operands (DVFS: MI355X_MICROARCH.md)."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.argv=["x","none"]; __file__="/root/repo/tools/ab_gemm256.py"
exec(open("/root/repo/tools/ab_gemm256.py").read().split('which = sys.argv[1]')[0])
M, hidden, K = 1024, 8192, 3072
for fill in ("random", "zero", "random"):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev) if fill == "random" else torch.zeros(M, K, device=dev)
    ws = [((torch.randn(2 * hidden, K, generator=g) * K ** -0.5) if fill == "random" else torch.zeros(2 * hidden, K)).to(torch.bfloat16).to(dev) for _ in range(4)]
    a2 = split(x); y = torch.empty(2, M, hidden, dtype=torch.bfloat16, device=dev)
    it = [0]
    def new():
        it[0] += 1
        check(L.mn_gemm256_swiglu_split(ptr(a2), K, a2.stride(0), ptr(ws[it[0] % 4]), K, None, ptr(y), hidden, y.stride(0), M, hidden, K, current_stream()), "x")
    med, mn = timeit(new, n=20, rounds=7)
    f = 2.0 * M * 2 * hidden * K * 2
    print(f"w12 M=1024 {fill}: {med*1e3:.1f} us {f/med/1e9:.0f} TF issued (min {f/mn/1e9:.0f})", flush=True)
