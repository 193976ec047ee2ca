"""gemm256 (256x256x64 tiles, 4-phase schedule) vs the simple one-barrier schedule vs the round-1 128x128 kernel:
correctness against an fp64 product and in-process interleaved timing (rule: same process, same data, random operands)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
from ming_univision_amd import ops
from ming_univision_amd._lib import lib, ptr, check, current_stream
L = lib()
L.mn_gemm256_tune = lambda sched: None   # the SCHED 0/1 arms were removed in round 3 (only the two-phase schedule is built)
L.mn_gemm_route256.argtypes = [ctypes.c_int]; L.mn_gemm_route256.restype = None
L.mn_gemm_route256(0)   # "round-1 kernel" below = the 128 x 128 kernel itself
EPI = {"bf16": 0, "f32": 2}
dev = "cuda"

def timeit(fn, n=10, rounds=5):
    return timeit_arms([fn], n, rounds)[0]


def timeit_arms(fns, n=10, rounds=7):
    """Interleaved rounds of all arms in one process (a first-measured arm otherwise pays the clock ramp): (median, min) per arm."""
    for fn in fns:
        for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for r in range(rounds):
        for i, fn in enumerate(fns):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n): fn()
            e.record(); torch.cuda.synchronize()
            ts[i].append(s.elapsed_time(e) / n)
    out = []
    for t in ts:
        t.sort()
        out.append((t[len(t) // 2], t[0]))
    return out

def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()       # [2, M, K]

def run_plain(M, N, K, epi="bf16", hilo=False, check_ref=True, nmat=1):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).to(dev)
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev) for _ in range(nmat)]
    w = ws[0]
    if hilo:
        a2 = split(x); a = a2[0]; lo_off = a2.stride(0)
        xr = a2[0].double() + a2[1].double()
    else:
        a = x.to(torch.bfloat16); lo_off = 0; xr = a.double()
    out = torch.empty(M, N, dtype=torch.float32 if epi == "f32" else torch.bfloat16, device=dev)
    it = [0]
    def new():
        it[0] += 1
        check(L.mn_gemm256(ptr(a), K, lo_off, ptr(ws[it[0] % nmat]), K, None, ptr(out), N, M, N, K, EPI[epi], current_stream()), "g256")
    res = {}
    ARMS = (2, 1)
    for sched in ARMS:
        L.mn_gemm256_tune(sched)
        it[0] = -1
        new(); torch.cuda.synchronize()
        err = float("nan")
        if check_ref:
            ref = xr @ w.double().T
            err = float((out.double() - ref).abs().max() / ref.abs().max())
        res[sched] = [None, err]
    def arm(sched):
        def f():
            L.mn_gemm256_tune(sched); new()
        return f
    for sched, t in zip(ARMS, timeit_arms([arm(a_) for a_ in ARMS])):
        res[sched][0] = t
    L.mn_gemm256_tune(2)
    old = None
    if not hilo:
        o2 = torch.empty_like(out)
        old = timeit(lambda: ops.gemm_bf16(a, w, None, epi, out=o2))
    elif epi == "f32":
        o2 = torch.empty_like(out)
        old = timeit(lambda: check(L.mn_gemm_bf16_hilo(ptr(a), K, lo_off, ptr(w), K, None, ptr(o2), N, M, N, K, current_stream()), "hilo"))
    f = 2.0 * M * N * K * (2 if hilo else 1)
    s = f"M={M} N={N} K={K} {epi}{' hilo' if hilo else ''}: "
    for sched in ARMS:
        (med, mn), err = res[sched]
        s += f"sched{sched} {med*1e3:.1f} us {f/med/1e9:.0f} TF (min {f/mn/1e9:.0f}) err {err:.1e} | "
    if old: s += f"round-1 kernel {old[0]*1e3:.1f} us {f/old[0]/1e9:.0f} TF"
    print(s, flush=True)

def run_swiglu(M, hidden, K, nmat=4):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev)
    ws = [(torch.randn(2 * hidden, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev) for _ in range(nmat)]
    b = (torch.randn(2 * hidden, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    a2 = split(x)
    y = torch.empty(2, M, hidden, dtype=torch.bfloat16, device=dev)
    it = [0]
    def new():
        it[0] += 1
        check(L.mn_gemm256_swiglu_split(ptr(a2), K, a2.stride(0), ptr(ws[it[0] % nmat]), K, ptr(b), ptr(y), hidden, y.stride(0), M, hidden, K, current_stream()), "swiglu")
    s = f"swiglu-split M={M} hidden={hidden} K={K}: "
    ARMS = (2, 1)
    errs = {}
    for sched in ARMS:
        L.mn_gemm256_tune(sched)
        it[0] = -1
        new(); torch.cuda.synchronize()
        xr = a2[0].double() + a2[1].double()
        r = xr @ ws[0].double().T + b.double()
        ref = torch.nn.functional.silu(r[:, :hidden]) * r[:, hidden:]
        got = y[0].double() + y[1].double()
        errs[sched] = float((got - ref).abs().max() / ref.abs().max())
    def arm(sched):
        def f_():
            L.mn_gemm256_tune(sched); new()
        return f_
    f = 2.0 * M * 2 * hidden * K * 2
    for sched, (med, mn) in zip(ARMS, timeit_arms([arm(a_) for a_ in ARMS])):
        s += f"sched{sched} {med*1e3:.1f} us {f/med/1e9:.0f} TF (min {f/mn/1e9:.0f}) err {errs[sched]:.1e} | "
    L.mn_gemm256_tune(2)
    print(s, flush=True)

def run_splitk(M, N, K, ks, nmat=4):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(M, K, generator=g).to(dev)
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev) for _ in range(nmat)]
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    a2 = split(x)
    P = torch.empty(ks + 1, M, N, dtype=torch.float32, device=dev)
    it = [0]; nz = [0]
    def new():
        it[0] += 1
        nz[0] = L.mn_gemm256_splitk(ptr(a2), K, a2.stride(0), ptr(ws[it[0] % nmat]), K, ptr(b), ptr(P), M, N, K, ks, current_stream())
        assert nz[0] >= 1
    it[0] = -1
    new(); torch.cuda.synchronize()
    ref = (a2[0].double() + a2[1].double()) @ ws[0].double().T + b.double()
    got = P[:nz[0]].double().sum(0)
    err = float((got - ref).abs().max() / ref.abs().max())
    med, mn = timeit(new)
    f = 2.0 * M * N * K * 2
    print(f"split-K M={M} N={N} K={K} hilo ksplit={ks} -> nz={nz[0]}: {med*1e3:.1f} us {f/med/1e9:.0f} TF (min {f/mn/1e9:.0f}) err {err:.1e}", flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "square"):
    for shp in [(512, 512, 512), (4096, 4096, 4096), (8192, 8192, 8192), (300, 1000, 192), (4160, 2304, 768), (16384, 4096, 1024), (16384, 1024, 4096)]:
        run_plain(*shp)
if which in ("all", "rf"):
    run_plain(4096, 116736, 3072, "f32", hilo=True, check_ref=False)
    run_plain(1024, 4096, 3072, "f32", hilo=True)
    for M in (256, 512, 1024):
        run_swiglu(M, 8192, 3072)
        for ks in (2, 4, 5, 8):
            run_splitk(M, 3072, 8192, ks)
