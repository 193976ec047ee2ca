"""Fixed cost per output tile of gemm256: same M x N, growing K (bf16 out and fp32 out)."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.argv=["x","none"]; __file__="/root/repo/tools/ab_gemm256.py"
exec(open("/root/repo/tools/ab_gemm256.py").read().split('which = sys.argv[1]')[0])
M, N = 16384, 4096
for epi in ("bf16", "f32"):
    for K in (128, 256, 512, 1024, 2048, 4096):
        g = torch.Generator().manual_seed(0)
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        out = torch.empty(M, N, dtype=torch.float32 if epi == "f32" else torch.bfloat16, device=dev)
        fn = lambda: check(L.mn_gemm256(ptr(a), K, 0, ptr(w), K, None, ptr(out), N, M, N, K, EPI[epi], current_stream()), "g")
        med, mn = timeit(fn, n=10, rounds=5)
        tiles = (M // 256) * (N // 256)
        print(f"{epi} K={K:5d}: {med*1e3:8.1f} us  -> {med*1e3/(tiles/256):6.1f} us per tile round ({K//64} K-tiles)", flush=True)
