"""The two expert launches of a 1- / 2-row decode step (ops.moe_experts: (row, expert) pairs with SwiGLU, then the down projection as
8 K-segments) at the 16B-A3B shapes under launch-plan overrides of the one-row kernel (mn_skinny_tune: rows per group, threads per
workgroup, workgroups per CU): us per layer for both launches, 3 distinct weight sets in turn."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
from ming_univision_amd import ops
from ming_univision_amd._lib import lib
L = lib()
L.mn_skinny_tune.argtypes = [ctypes.c_int] * 3; L.mn_skinny_tune.restype = None
g = torch.Generator().manual_seed(0)
E, S, I, H, top = 64, 2, 1408, 2048, 6
sets = [((torch.randn(E + S, 2 * I, H, device="cuda") * H ** -0.5).to(torch.bfloat16), (torch.randn(E + S, H, I, device="cuda") * I ** -0.5).to(torch.bfloat16))
        for _ in range(3)]
def timed(fn, n=30):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
for M in (1, 2):
    xn = torch.randn(M, H, generator=g).cuda(); res = torch.randn(M, H, generator=g).cuda()
    idx = torch.stack([torch.cat((torch.randperm(E, generator=g)[:top], torch.tensor([E, E + 1]))) for _ in range(M)]).to(torch.int32).cuda()
    w = torch.cat((torch.rand(M, top, generator=g), torch.ones(M, S)), 1).cuda()
    for tune in ((0, 0, 0), (0, 768, 1), (0, 1024, 1), (1, 1024, 1), (1, 768, 1)):
        L.mn_skinny_tune(*tune)
        t = timed(lambda i: ops.moe_experts(xn, idx, w, sets[i % 3][0], sets[i % 3][1], res))
        print(f"rows {M} tune (R, nt, bpc) = {tune}: {t:6.1f} us per layer (gate_up + down)", flush=True)
L.mn_skinny_tune(0, 0, 0)
