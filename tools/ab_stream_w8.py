"""fp8 vs bf16 weight-streaming launches, in-process interleaved (HIP events): RF w12 / w3 dense at several row counts, the
grouped expert launches of one 16B-A3B layer, and the fp8 K-slice kernel's ring depth (1 vs 2 chunks in flight per wave)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.devlib  # noqa: F401
from ming_univision_amd import ops
from ming_univision_amd._lib import lib, ptr, current_stream
L = lib()
L.mn_stream_tune_w8.argtypes = [ctypes.c_int]; L.mn_stream_tune_w8.restype = None


def timed(fn, n=24, warm=3):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


def dense(N, K, rows_list, nw=6):
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
    qs = [ops.quant_fp8_rows(w) for w in ws]
    for M in rows_list:
        Y = (torch.randn(2 * M, K, device="cuda") * 0.5).to(torch.bfloat16)
        P = torch.empty(64 * M * N, device="cuda")
        res = {}
        for r in range(5):
            res.setdefault("bf16", []).append(timed(lambda i: L.mn_stream_mfma(ptr(Y), ptr(ws[i % nw]), ptr(P), M, N, K, current_stream())))
            for d in (1, 2):
                L.mn_stream_tune_w8(d)
                res.setdefault("fp8 d%d" % d, []).append(timed(
                    lambda i: L.mn_stream_mfma_w8(ptr(Y), ptr(qs[i % nw][0]), ptr(qs[i % nw][1]), ptr(P), M, N, K, current_stream())))
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print(f"dense N={N} K={K} rows={M}: " + "  ".join(
            f"{k}: {t:.1f} us ({N * K * (2 if k == 'bf16' else 1) / t / 1e6:.2f} TB/s)" for k, t in med.items()), flush=True)
    L.mn_stream_tune_w8(2)


def grouped(M, E=64, S=2, top=6):
    g = torch.Generator().manual_seed(0)
    for name, N, K, gather in (("gate_up", 2816, 2048, True), ("down", 2048, 1408, False)):
        idx = torch.stack([torch.randperm(E, generator=g)[:top] for _ in range(M)])
        cnt = torch.bincount(idx.flatten(), minlength=E).tolist() + [M] * S
        off = torch.tensor([0] + list(torch.tensor(cnt).cumsum(0)), dtype=torch.int32)
        total = int(off[-1])
        xrows = torch.randint(0, M, (total,), generator=g, dtype=torch.int32).cuda() if gather else None
        nx = M if gather else total
        Y = (torch.randn(2 * nx, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
        W = [(torch.randn(E + S, N, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(3)]
        Q = [ops.quant_fp8_rows(w) for w in W]
        P = torch.empty(4 * total * N, device="cuda")
        offd = off.cuda()
        live = sum(1 for c in cnt if c > 0)
        res = {}
        for r in range(5):
            res.setdefault("bf16", []).append(timed(lambda i: L.mn_stream_mfma_grouped(
                ptr(Y), nx, ptr(W[i % 3]), N * K, ptr(P), total, ptr(offd), ptr(xrows), E + S, M, N, K, current_stream()), n=9, warm=2))
            res.setdefault("fp8", []).append(timed(lambda i: L.mn_stream_mfma_grouped_w8(
                ptr(Y), nx, ptr(Q[i % 3][0]), N * K, ptr(Q[i % 3][1]), N, ptr(P), total, ptr(offd), ptr(xrows), E + S, M, N, K,
                current_stream()), n=9, warm=2))
        med = {k: sorted(v)[2] for k, v in res.items()}
        print(f"experts {name} rows={M} ({live} live experts): " + "  ".join(
            f"{k}: {t:.0f} us ({live * N * K * (2 if k == 'bf16' else 1) / t / 1e6:.2f} TB/s)" for k, t in med.items()), flush=True)


if __name__ == "__main__":
    dense(2 * 8192, 3072, [2, 3, 16, 32, 48, 64])
    dense(3072, 8192, [2, 3, 16, 32, 48, 64])
    for M in (1, 2, 3, 16, 64):
        grouped(M)
