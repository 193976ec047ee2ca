"""Round-1 32-key-tile attention kernels against flash_prefill.hip (64-key tiles, transposing LDS read): MingTok shapes
(hd 64: 64 x 257 causal / 64 x 1024 full / 1 x 4096 full) and the LLM prompt (hd 128 GQA 16:4, T = 1058)."""
import sys, os, ctypes, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd._lib import lib, ptr, check, current_stream
L = lib()
L.mn_attn_tune.argtypes = [ctypes.c_int]; L.mn_attn_tune.restype = None
def ev(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
g = torch.Generator(device="cuda").manual_seed(0)
for (B, T, nh, causal) in ((64, 257, 16, 1), (64, 1024, 16, 0), (64, 65, 12, 0), (1, 4096, 16, 0), (8, 1025, 12, 0)):
    qkv = (torch.randn(B, T, 3, nh, 64, device="cuda", generator=g)).to(torch.bfloat16)
    outs = []
    for flash in (0, 1):
        L.mn_attn_tune(flash)
        out = torch.empty(B, T, nh * 64, dtype=torch.bfloat16, device="cuda")
        f = lambda: check(L.mn_attn_prefill_hd64(ptr(qkv), ptr(out), B, T, nh, causal, current_stream()), "attn")
        t = ev(f)
        fl = 4.0 * B * nh * T * T * 64 * (0.5 if causal else 1.0)
        outs.append(out.float())
        print(f"hd64 B={B} T={T} nh={nh} causal={causal} flash={flash}: {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s", flush=True)
    print("   max |diff| =", float((outs[0] - outs[1]).abs().max()), " ref max", float(outs[0].abs().max()))
nq, nkv, t_max = 16, 4, 1152
for T in (1058, 300):
    q = (torch.randn(T, nq, 128, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    kv = torch.randn(2, nkv, t_max, 128, device="cuda", generator=g)
    outs = []
    for flash in (0, 1):
        L.mn_attn_tune(flash)
        out = torch.empty(T, nq * 128, dtype=torch.bfloat16, device="cuda")
        f = lambda: check(L.mn_attn_prefill_gqa_hd128(ptr(q), ptr(kv), t_max, nq, nkv, 0, T, None, ptr(out), current_stream()), "attn")
        t = ev(f)
        fl = 4.0 * nq * T * T * 128 * 0.5
        outs.append(out.float())
        print(f"hd128 GQA T={T} flash={flash}: {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s", flush=True)
    print("   max |diff| =", float((outs[0] - outs[1]).abs().max()), " ref max", float(outs[0].abs().max()))
L.mn_attn_tune(1)
