"""One attention configuration, a few launches (for rocprofv3 --pmc): argv = hd64|hd128 flash(0/1)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd._lib import lib, ptr, check, current_stream
L = lib()
which = sys.argv[1]      # (argv[2] selected the round-1 32-key kernels until they were removed in round 3)
g = torch.Generator(device="cuda").manual_seed(0)
if which == "hd64":
    B, T, nh = 64, 1024, 16
    qkv = torch.randn(B, T, 3, nh, 64, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(B, T, nh * 64, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        check(L.mn_attn_prefill_hd64(ptr(qkv), ptr(out), B, T, nh, 0, current_stream()), "attn")
elif which == "hd128x7":
    nq, nkv, t_max, T, S = 16, 4, 1152, 1058, 7
    q = (torch.randn(S * T, nq, 128, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    kv = torch.randn(S, 2, nkv, t_max, 128, device="cuda", generator=g)
    tab = torch.tensor([[i, i * T, T] for i in range(S)], dtype=torch.int32).cuda()
    out = torch.empty(S * T, nq * 128, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        check(L.mn_flash_prefill_gqa_hd128(ptr(q), ptr(kv), t_max, nq, nkv, 0, ptr(tab), S, T, None, 0, ptr(out), current_stream()), "attn")
else:
    nq, nkv, t_max, T = 16, 4, 1152, 1058
    q = (torch.randn(T, nq, 128, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    kv = torch.randn(2, nkv, t_max, 128, device="cuda", generator=g)
    out = torch.empty(T, nq * 128, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        check(L.mn_attn_prefill_gqa_hd128(ptr(q), ptr(kv), t_max, nq, nkv, 0, T, None, ptr(out), current_stream()), "attn")
torch.cuda.synchronize()
